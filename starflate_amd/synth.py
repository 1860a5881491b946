"""Seeded synthetic corpora standing in for the corpora BASELINE.json names.

The Canterbury / Silesia / enwik files are not in the image and cannot be
fetched, so every config uses a generator:

  gen_text(n, seed)    enwik-like text: Zipf-distributed draws from a synthetic
                       65,536-word vocabulary with English-like letter
                       frequencies, punctuation, line breaks, ~5 % wiki markup.
  gen_random(n, seed)  high-entropy bytes (pre-compressed data stand-in).
  gen_mixed(n, seed)   "Silesia-like tar": stripes of text / repetitive XML /
                       binary records / smooth image-like bytes / random.

Only numpy `integers` / `random` primitives are used so the bytes depend on the
seed alone.
"""
import numpy as np

_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
_LETTER_W = np.array(
    [12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4,
     2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07]
)
_SEPS = [b" "] * 80 + [b", "] * 6 + [b". "] * 5 + [b".\n"] * 2 + [b"\n\n"] + [
    b" [[", b"]] ", b" <ref>", b"</ref> ", b" {{", b"}} ", b" &quot;", b"&quot; ", b" 19", b" 20"]
_VOCAB = 65536


def _ragged_gather(blob, starts, lens):
    """Concatenate blob[starts[k] : starts[k]+lens[k]] for all k."""
    total = int(lens.sum())
    ends = np.cumsum(lens)
    begin = ends - lens
    idx = np.arange(total, dtype=np.int64) - np.repeat(begin - starts, lens)
    return blob[idx]


_NPHRASE = 8192


class _TextModel:
    """Unit table = 65,536 words + 8,192 multi-word phrases (both Zipf-ranked) + separators."""

    def __init__(self, seed):
        rng = np.random.Generator(np.random.PCG64(seed ^ 0x5F17A7E))
        wl = np.minimum(2 + np.floor(np.log1p(-rng.random(_VOCAB)) / np.log(1 - 0.25)), 16).astype(np.int64)
        # the most frequent ranks are short function words, like natural language
        wl[:64] = rng.integers(1, 5, 64)
        wl[64:512] = np.minimum(wl[64:512], rng.integers(2, 8, 448))
        cdf = np.cumsum(_LETTER_W / _LETTER_W.sum())
        letters = _LETTERS[np.searchsorted(cdf, rng.random(int(wl.sum())), side="right").clip(0, 25)]
        wstart = np.cumsum(wl) - wl
        w = 1.0 / np.power(np.arange(1, _VOCAB + 1, dtype=np.float64), 1.1)
        self.word_cdf = np.cumsum(w / w.sum())
        # phrases: 2..5 words joined by single spaces
        pw = rng.integers(2, 6, _NPHRASE)
        ids = np.searchsorted(self.word_cdf, rng.random(int(pw.sum())), side="right").clip(0, _VOCAB - 1)
        space = letters.size  # index of a ' ' appended below
        blob0 = np.concatenate([letters, np.frombuffer(b" ", dtype=np.uint8)])
        starts2 = np.empty(2 * ids.size, dtype=np.int64)
        lens2 = np.empty(2 * ids.size, dtype=np.int64)
        starts2[0::2] = wstart[ids]
        lens2[0::2] = wl[ids]
        starts2[1::2] = space
        lens2[1::2] = 1
        last = np.cumsum(pw) * 2 - 1  # trailing space of each phrase: drop it
        lens2[last] = 0
        phrase_bytes = _ragged_gather(blob0, starts2, lens2)
        per_item = np.add.reduceat(lens2, np.concatenate([[0], np.cumsum(pw)[:-1] * 2]))
        sep_blob = np.frombuffer(b"".join(_SEPS), dtype=np.uint8)
        sep_len = np.array([len(s) for s in _SEPS], dtype=np.int64)
        self.blob = np.concatenate([letters, phrase_bytes, sep_blob])
        lens = np.concatenate([wl, per_item, sep_len])
        self.lens = lens
        self.starts = (np.cumsum(lens) - lens).astype(np.int64)
        pz = 1.0 / np.power(np.arange(1, _NPHRASE + 1, dtype=np.float64), 1.0)
        self.phrase_cdf = np.cumsum(pz / pz.sum())
        self.nsep = len(_SEPS)

    def piece(self, n, rng):
        chunks = []
        have = 0
        while have < n:
            k = max(1024, int((n - have) / 6.0) + 1024)
            words = np.searchsorted(self.word_cdf, rng.random(k), side="right").clip(0, _VOCAB - 1)
            phr = _VOCAB + np.searchsorted(self.phrase_cdf, rng.random(k), side="right").clip(0, _NPHRASE - 1)
            units = np.where(rng.random(k) < 0.30, phr, words)
            seps = _VOCAB + _NPHRASE + rng.integers(0, self.nsep, k)
            items = np.empty(2 * k, dtype=np.int64)
            items[0::2] = units
            items[1::2] = seps
            c = _ragged_gather(self.blob, self.starts[items], self.lens[items])
            chunks.append(c)
            have += c.size
        return np.concatenate(chunks)[:n]


_MODELS = {}


def gen_text(n, seed=3, piece_bytes=1 << 24):
    """n bytes of enwik-like synthetic text (uint8 ndarray)."""
    model = _MODELS.get(seed)
    if model is None:
        model = _MODELS[seed] = _TextModel(seed)
    out = np.empty(n, dtype=np.uint8)
    pos, k = 0, 0
    while pos < n:
        m = min(piece_bytes, n - pos)
        rng = np.random.Generator(np.random.PCG64([seed, k]))
        out[pos:pos + m] = model.piece(m, rng)
        pos += m
        k += 1
    return out


def gen_random(n, seed=5):
    """n high-entropy bytes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, n, dtype=np.uint8)


def _gen_xml(n, rng):
    tags = [b"<row id=\"", b"\" name=\"", b"\" value=\"", b"\"/>\n", b"  <item>", b"</item>\n",
            b"<entry type=\"int\">", b"</entry>\n", b"    return self.", b"(self, x):\n", b"if (", b") {\n", b"}\n"]
    blob = np.frombuffer(b"".join(tags) + b"0123456789abcdef", dtype=np.uint8)
    lens = np.array([len(t) for t in tags] + [1] * 16, dtype=np.int64)
    starts = np.cumsum(lens) - lens
    k = n // 4 + 64
    items = np.where(rng.random(k) < 0.55, rng.integers(0, len(tags), k), len(tags) + rng.integers(0, 16, k))
    return _ragged_gather(blob, starts[items], lens[items])[:n]


def _gen_records(n, rng):
    k = n // 16 + 1
    rec = np.zeros((k, 4), dtype=np.uint32)
    rec[:, 0] = np.arange(k, dtype=np.uint32)
    rec[:, 1] = (1000 + np.cumsum(rng.integers(-3, 4, k))).astype(np.uint32)
    rec[:, 2] = rng.integers(0, 16, k).astype(np.uint32)
    rec[:, 3] = 0xDEADBEEF
    return rec.view(np.uint8).reshape(-1)[:n]


def _gen_image(n, rng):
    walk = np.cumsum(rng.integers(-2, 3, n)) // 2
    return ((walk + rng.integers(0, 3, n)) & 0xFF).astype(np.uint8)


def gen_mixed(n, seed=4, stripe=1 << 20):
    """Silesia-like mix: per 10 stripes 4 text, 2 XML/source, 2 records, 1 image, 1 random."""
    kinds = [0, 1, 2, 0, 3, 0, 1, 2, 0, 4]
    out = np.empty(n, dtype=np.uint8)
    pos, k = 0, 0
    while pos < n:
        m = min(stripe, n - pos)
        rng = np.random.Generator(np.random.PCG64([seed, k]))
        kind = kinds[k % len(kinds)]
        if kind == 0:
            model = _MODELS.get(seed) or _MODELS.setdefault(seed, _TextModel(seed))
            out[pos:pos + m] = model.piece(m, rng)
        elif kind == 1:
            out[pos:pos + m] = _gen_xml(m, rng)
        elif kind == 2:
            out[pos:pos + m] = _gen_records(m, rng)
        elif kind == 3:
            out[pos:pos + m] = _gen_image(m, rng)
        else:
            out[pos:pos + m] = rng.integers(0, 256, m, dtype=np.uint8)
        pos += m
        k += 1
    return out


def gen_text_torch(n, seed=3, device="cuda", piece_bytes=1 << 26):
    """Same text model as gen_text, sampled on the GPU with torch's generator
    (bench-sized inputs: 1 GiB takes well under a second per piece instead of minutes).
    The bytes differ from gen_text(n, seed) -- only the distribution is shared."""
    import torch

    model = _MODELS.get(seed)
    if model is None:
        model = _MODELS[seed] = _TextModel(seed)
    dev = torch.device(device)
    blob = torch.from_numpy(model.blob).to(dev)
    starts = torch.from_numpy(model.starts).to(dev)
    lens = torch.from_numpy(model.lens).to(dev)
    wcdf = torch.from_numpy(model.word_cdf).to(dev)
    pcdf = torch.from_numpy(model.phrase_cdf).to(dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    pos, k = 0, 0
    while pos < n:
        m = min(piece_bytes, n - pos)
        g.manual_seed((seed << 20) + k)
        have, parts = 0, []
        while have < m:
            cnt = max(1024, int((m - have) / 6.0) + 1024)
            words = torch.searchsorted(wcdf, torch.rand(cnt, generator=g, device=dev, dtype=torch.float64)).clamp_(0, _VOCAB - 1)
            phr = _VOCAB + torch.searchsorted(pcdf, torch.rand(cnt, generator=g, device=dev, dtype=torch.float64)).clamp_(0, _NPHRASE - 1)
            units = torch.where(torch.rand(cnt, generator=g, device=dev) < 0.30, phr, words)
            seps = _VOCAB + _NPHRASE + torch.randint(0, model.nsep, (cnt,), generator=g, device=dev)
            items = torch.stack([units, seps], dim=1).reshape(-1)
            il = lens[items]
            total = int(il.sum())
            begin = torch.cumsum(il, 0) - il
            idx = torch.arange(total, device=dev) - torch.repeat_interleave(begin - starts[items], il)
            parts.append(blob[idx])
            have += total
        piece = torch.cat(parts)[:m]
        out[pos:pos + m] = piece
        pos += m
        k += 1
    return out
