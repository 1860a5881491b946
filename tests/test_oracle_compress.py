"""The encoder specification (oracle/sf_oracle.c part 2) on CPU: every stream it emits must
pass the oracle's restatement of the reference decoder AND zlib inflate, for every strategy,
edge-case input and parameter set; plus the emitter-contract properties of SURVEY.md
Appendix A (complete length-limited codes, per-sequence RLE, byte-aligned non-final shards)."""
import zlib

import numpy as np
import pytest

import oracle_lib as O
from starflate_amd import synth

CHUNK = 32768


def _cases(starfleet):
    rng = np.random.default_rng(11)
    text = synth.gen_text(150_000, seed=2)
    return {
        "empty": np.zeros(0, np.uint8),
        "one": np.array([0], np.uint8),
        "two": np.frombuffer(b"ab", np.uint8),
        "four_same": np.frombuffer(b"aaaa", np.uint8),
        "run_300": np.full(300, 7, np.uint8),
        "zeros_chunk": np.zeros(CHUNK, np.uint8),
        "zeros_ragged": np.zeros(2 * CHUNK + 5, np.uint8),
        "text": text,
        "text_chunk_pm1": text[: CHUNK + 1],
        "html": np.frombuffer(starfleet, np.uint8),
        "random": rng.integers(0, 256, CHUNK + 17, dtype=np.uint8),
        "all_bytes": np.tile(np.arange(256, dtype=np.uint8), 200),
        "two_symbols": rng.integers(0, 2, 50_000, dtype=np.uint8),
        "fib_freqs": np.repeat(np.arange(24, dtype=np.uint8), [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765, 10946, 17711, 28657, 46368])[:CHUNK],
        "mixed": synth.gen_mixed(600_000, seed=4, stripe=1 << 15),
    }


def _check(stream, data):
    st, w, out = O.decompress(stream, data.size)
    assert st == 0 and w == data.size and np.array_equal(out, data)
    assert zlib.decompress(bytes(stream), -15) == data.tobytes()


@pytest.mark.parametrize("strategy", [0, 1, 2, 3])
def test_roundtrip_every_strategy(starfleet, strategy):
    for name, data in _cases(starfleet).items():
        s = O.compress(data, O.default_params(strategy=strategy))
        _check(s, data)
        assert s.size <= O.lib().sfo_compress_bound(data.size, O.default_params()), name


@pytest.mark.parametrize("kw", [
    dict(lazy=0), dict(step=256), dict(step=4096), dict(hash_bits=10), dict(hash_bits=15),
    dict(region_bytes=64), dict(region_bytes=2048), dict(region_bytes=32768), dict(cap=0), dict(cap=8),
    dict(min_match=3), dict(depth=2), dict(depth=3, long_hash_bytes=7), dict(use_near=0),
    dict(chunk_bytes=4096, region_bytes=512), dict(chain_depth=8),
])
def test_roundtrip_parameter_space(starfleet, kw):
    for name in ("text", "html", "zeros_ragged", "mixed"):
        data = _cases(starfleet)[name]
        _check(O.compress(data, O.default_params(**kw)), data)


def test_auto_never_exceeds_stored(starfleet):
    for name, data in _cases(starfleet).items():
        auto = O.compress(data, O.default_params(strategy=0))
        stored = O.compress(data, O.default_params(strategy=1))
        assert auto.size <= stored.size, name
    nchunks = 3
    rnd = np.random.default_rng(3).integers(0, 256, nchunks * CHUNK, dtype=np.uint8)
    assert O.compress(rnd).size == rnd.size + 5 * nchunks  # stored fast path: 5 bytes per block


def test_non_final_shards_concatenate(starfleet):
    """Every shard but the last ends byte-aligned and non-final (final_stream=0), so plain
    concatenation is one valid stream -- the multi-GPU contract."""
    data = np.frombuffer(starfleet, np.uint8)
    cuts = [0, CHUNK, 3 * CHUNK, data.size]
    parts = [O.compress(data[a:b], O.default_params(final_stream=int(b == data.size))) for a, b in zip(cuts, cuts[1:])]
    _check(np.concatenate(parts), data)
    whole = O.compress(data)
    assert sum(p.size for p in parts) == whole.size and np.array_equal(np.concatenate(parts), whole)


def test_build_lengths_properties():
    rng = np.random.default_rng(5)
    for trial in range(300):
        n = int(rng.choice([19, 30, 286]))
        maxbits = 7 if n == 19 else 15
        kind = trial % 4
        if kind == 0:
            f = rng.integers(0, 50, n)
        elif kind == 1:
            f = (rng.pareto(0.7, n) * 3).astype(np.int64).clip(0, 32768)
        elif kind == 2:
            f = np.zeros(n, np.int64)
            f[rng.choice(n, size=int(rng.integers(0, 4)), replace=False)] = rng.integers(1, 100)
        else:
            fib = [1, 1]
            while len(fib) < n:
                fib.append(min(fib[-1] + fib[-2], 32768))
            f = np.array(fib[:n]) * (rng.random(n) < 0.9)
        lens = O.build_lengths(f, maxbits)
        used = f > 0
        assert np.all(lens[~used] == 0) and np.all(lens[used] >= 1) and lens.max(initial=0) <= maxbits
        m = int(used.sum())
        kraft = sum(2.0 ** -int(l) for l in lens[used])
        if m >= 2:
            assert abs(kraft - 1.0) < 1e-12, (trial, kraft)  # complete: zlib rejects incomplete CL codes
        elif m == 1:
            assert lens[used][0] == 1
        # rarer symbols never get shorter codes
        order = np.argsort(f[used], kind="stable")
        assert np.all(np.diff(lens[used][order].astype(int)) <= 0)


def test_huffman_cost_is_optimal_when_unconstrained():
    """Without the length limit biting, the code must cost exactly what a textbook Huffman code costs."""
    import heapq

    rng = np.random.default_rng(9)
    for _ in range(50):
        f = rng.integers(1, 200, 40)
        lens = O.build_lengths(np.concatenate([f, np.zeros(246, np.int64)]), 15)
        h = [(int(x), i) for i, x in enumerate(f)]
        heapq.heapify(h)
        cost, k = 0, len(f)
        while len(h) > 1:
            a, b = heapq.heappop(h), heapq.heappop(h)
            cost += a[0] + b[0]
            heapq.heappush(h, (a[0] + b[0], k))
            k += 1
        assert int((lens[:40].astype(np.int64) * f).sum()) == cost


def test_dynamic_header_obeys_decoder_hazards(starfleet):
    """Parse the dynamic header of every chunk the spec emits: the HLIT and HDIST length
    sequences must each decode on their own (no run crossing over, no leading 16, no overshoot).
    sfo_decompress returns SFO_ERROR(1) on exactly those, so a clean status 0 is the check;
    here the RLE items are additionally counted from the plan."""
    data = np.frombuffer(starfleet, np.uint8)
    p = O.default_params(strategy=3)
    for c in range(0, data.size, CHUNK):
        d = data[c:c + CHUNK]
        ln, ds = O.match_chunk(d, p)
        t, nt = O.parse_chunk(d, p, ln, ds)
        ll, dd = O.histogram(t, nt, p.region_bytes)
        pl = O.plan_chunk(ll, dd, d.size, False, p)
        assert pl.btype == 2 and 17 <= pl.header_bits <= 4492
        assert max(pl.ll_lens) <= 15 and max(pl.d_lens) <= 15
        assert pl.ll_lens[256] >= 1 and pl.ll_lens[286] == 0 and pl.ll_lens[287] == 0
        assert pl.d_lens[30] == 0 and pl.d_lens[31] == 0


def test_ratio_floor_vs_zlib6(starfleet):
    """Guard rail on match quality: the block-parallel parse must stay within 25 % of zlib -6."""
    for data in (synth.gen_text(1 << 20, seed=3), np.frombuffer(starfleet, np.uint8)):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        z = len(co.compress(data.tobytes())) + len(co.flush())
        assert O.compress(data).size <= 1.33 * z


def test_checksums_match_zlib(starfleet):
    """The oracle's bit-serial CRC-32 / Adler-32 and their combine rules against zlib's (golden: zlib itself)."""
    rng = np.random.default_rng(21)
    for n in (0, 1, 2, 255, 5552, 5553, 32768, 65521, 100_003):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        assert O.crc32(d) == zlib.crc32(d.tobytes()) and O.adler32(d) == zlib.adler32(d.tobytes())
    ff = np.full(70_000, 255, np.uint8)  # Adler worst case for overflow
    assert O.adler32(ff) == zlib.adler32(ff.tobytes())
    d = np.frombuffer(starfleet, np.uint8)
    for cut in (0, 1, 32768, 99_999, d.size):
        a, b = d[:cut].tobytes(), d[cut:].tobytes()
        assert O.lib().sfo_crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(d.tobytes())
        assert O.lib().sfo_adler32_combine(zlib.adler32(a), zlib.adler32(b), len(b)) == zlib.adler32(d.tobytes())


@pytest.mark.parametrize("container", [1, 2])
def test_container_wrappers(starfleet, container):
    """zlib (RFC 1950) and gzip (RFC 1952) wrappers: the body is the raw stream unchanged, and zlib's
    own wrapper-checking inflate (wbits 15 / 31) accepts the whole thing (header, checksum, ISIZE)."""
    import gzip

    for name, data in _cases(starfleet).items():
        raw = O.compress(data)
        s = O.compress(data, O.default_params(container=container))
        h, t = (2, 4) if container == 1 else (10, 8)
        assert s.size == raw.size + h + t and np.array_equal(s[h:-t], raw), name
        assert zlib.decompress(bytes(s), 15 if container == 1 else 31) == data.tobytes(), name
        if container == 2:
            assert gzip.decompress(bytes(s)) == data.tobytes()
        assert s.size <= O.lib().sfo_compress_bound(data.size, O.default_params())
    with pytest.raises(RuntimeError):  # a non-final shard cannot carry a trailer
        O.compress(b"abc", O.default_params(container=container, final_stream=0))


def test_runs_are_coded_at_distance_one():
    """A long match whose bytes all equal the byte before it is coded at distance 1 with the run's length (an
    overlapping copy, /root/reference/src/decompress.cpp:388-398): a chunk of one byte value is 2 matches per 512-byte
    region, all at distance 1, and still round-trips; a periodic input (not a run) keeps its hash-table distances."""
    zeros = np.zeros(3 * 32768 + 100, np.uint8)
    p = O.default_params(strip_bytes=65536)
    toks = O.strip_tokens(zeros[:65536], p)[0]
    m = toks[(toks & 0x80000000) != 0]
    assert m.size >= 2 * (65536 // 512) - 2 and np.all((m & 0x7FFF) == 0)          # dist - 1 == 0
    assert np.sum(((m >> 16) & 0xFF) == 255) >= 65536 // 512 - 1                      # a 258 in every region
    s = O.compress(zeros, p)
    st, w, back = O.decompress(s, zeros.size)
    assert st == 0 and w == zeros.size and np.array_equal(back, zeros)
    assert zlib.decompress(s.tobytes(), -15) == zeros.tobytes()
    off = O.compress(zeros, O.default_params(strip_bytes=65536, run_dist1=0))
    assert s.size < 0.6 * off.size
    line = np.tile(np.arange(61, dtype=np.uint8) + 40, 2000)[:65536]
    tl = O.strip_tokens(line, p)[0]
    ml = tl[(tl & 0x80000000) != 0]
    assert np.all(((ml & 0x7FFF) + 1) % 61 == 0)
    # a run shorter than the threshold, or one that the match does not cover entirely, is left alone
    mix = np.concatenate([np.arange(200, dtype=np.uint8), np.full(40, 7, np.uint8), np.arange(200, dtype=np.uint8),
                          np.full(40, 7, np.uint8), np.arange(50, dtype=np.uint8)])
    sm = O.compress(mix, O.default_params())
    assert zlib.decompress(sm.tobytes(), -15) == mix.tobytes()


def test_stride2_searches_even_positions_and_inherits_from_the_successor(starfleet):
    """stride2 (every effort but thorough): only even strip positions are searched; an odd position has a match only as its
    successor's, one byte longer at the same distance, with its own byte matching too -- inside the step and the region."""
    data = np.frombuffer(starfleet, np.uint8)[:32768].copy()
    p2 = O.default_params()
    assert p2.stride2 == 1 and p2.step == 1024 and p2.hash_bits == 13
    ln, ds = O.match_chunk(data, p2)
    full_ln, full_ds = O.match_chunk(data, O.default_params(stride2=0))
    # even positions: exactly what the full search finds there
    assert np.array_equal(ln[0::2], full_ln[0::2]) and np.array_equal(ds[0::2][ln[0::2] > 0], full_ds[0::2][full_ln[0::2] > 0])
    odd = np.flatnonzero(ln[1::2] > 0) * 2 + 1
    assert odd.size > 100
    for i in odd[:2000]:
        i = int(i)
        assert ln[i + 1] > 0 and ds[i] == ds[i + 1]                      # the successor's match ...
        assert ln[i] == min(int(ln[i + 1]) + 1, p2.cap)                  # ... one byte longer, capped
        assert data[i] == data[i - int(ds[i])] and int(ds[i]) <= i       # its own byte fits, inside the strip
        assert (i + 1) % p2.step != 0 and (i + 1) % p2.region_bytes != 0  # same step, same parse region
    # every effort round-trips; thorough (all positions, steps of 512) is never larger on this text
    sizes = {}
    for name, kw in (("max", dict(stride2=0, step=512, hash_bits=12, long_hash_bytes=7)), ("thorough", dict(stride2=0, step=512)), ("default", {}), ("fast", dict(depth=1)), ("fastest", dict(depth=1, use_near=0))):
        s = O.compress(np.frombuffer(starfleet, np.uint8), O.default_params(**kw))
        _check(s, np.frombuffer(starfleet, np.uint8))
        sizes[name] = s.size
    assert sizes["max"] <= sizes["thorough"] <= sizes["default"] <= sizes["fast"] <= sizes["fastest"]


def test_hash_chains_are_exact_and_most_recent_first(starfleet):
    """chain_depth > 0 (SFH_EFFORT_BEST / _ULTRA): the match of a position is what a brute-force walk over the most recent
    earlier positions with the same 13-bit hash finds -- at most chain_depth of them, none farther back than 32 KiB, every
    one compared up to `cap` bytes, the longest winning and the nearest on ties -- and deeper chains never compress worse."""
    data = np.frombuffer(starfleet, np.uint8)[:32768].copy()
    p = O.default_params(chain_depth=8)
    ln, ds = O.match_chunk(data, p)
    v = data.astype(np.uint32)
    w = v[:-3] | (v[1:-2] << 8) | (v[2:-1] << 16) | (v[3:] << 24)
    h = ((w.astype(np.uint64) * 2654435761) & 0xFFFFFFFF) >> (32 - p.hash_bits)
    seen = {}
    checked = 0
    for i in range(data.size - 3):
        hist = seen.setdefault(int(h[i]), [])
        if i % 7 == 0:  # a sample of positions, every one of them inserted
            maxlen = min(data.size - i, 258, (i // p.region_bytes + 1) * p.region_bytes - i, p.cap)
            best, bd = 0, 0
            for c in hist[-1:-9:-1]:
                if i - c > 32768:
                    break
                ln_c = 0
                while ln_c < maxlen and data[i + ln_c] == data[c + ln_c]:
                    ln_c += 1
                if ln_c > best:
                    best, bd = ln_c, i - c
                if best == p.cap:
                    break
            if best == 4 and bd > p.far4_dist:
                best = 0
            if best >= 4:
                assert (int(ln[i]), int(ds[i])) == (best, bd), i
            else:
                assert ln[i] == 0, i
            checked += 1
        hist.append(i)
    assert checked > 4000
    whole = np.frombuffer(starfleet, np.uint8)
    sizes = []
    for depth in (0, 4, 8, 16):
        s = O.compress(whole, O.default_params(chain_depth=depth))
        _check(s, whole)
        sizes.append(s.size)
    assert sizes[3] <= sizes[2] <= sizes[1] and sizes[2] < sizes[0]


def test_recent_buckets_hold_exact_recency(starfleet):
    """recent = 1 (SFH_EFFORT_RECENT / _RECENT_ALL): a position's candidates are its exact predecessor(s) with the same hash
    -- near_depth of them as far as the links reach (link_steps steps, the current one included) -- plus the bucket's lo and
    hi as they stood before its step: lo the latest earlier position with the hash, hi what lo was before the most recent
    step that inserted the hash.  A brute-force model of exactly that must find the same match at every searched position,
    for both search patterns; and the specification never compresses worse than the step tables it replaces."""
    data = np.concatenate([np.frombuffer(starfleet, np.uint8)[:40000], np.zeros(3000, np.uint8), synth.gen_text(22536, seed=5)])
    v = data.astype(np.uint32)
    w = v[:-3] | (v[1:-2] << 8) | (v[2:-1] << 16) | (v[3:] << 24)
    for kw in (dict(near_depth=1, link_steps=1), dict(near_depth=3, link_steps=2), dict(near_depth=1, link_steps=1, stride2=0, step=512),
               dict(near_depth=2, link_steps=4, stride2=0, step=512)):
        p = O.default_params(recent=1, strip_bytes=65536 * 2, chunk_bytes=32768, **kw)
        W, R = p.step, p.region_bytes
        h = ((w.astype(np.uint64) * 2654435761) & 0xFFFFFFFF) >> (32 - p.hash_bits)
        n = data.size
        # match_chunk treats its input as one strip: use a strip-sized prefix
        strip = data[:65536]
        ln, ds = O.match_chunk(strip, p)
        n = strip.size
        lo, hi, prev = {}, {}, {}
        checked = 0
        for s0 in range(0, n, W):
            e = min(s0 + W, n)
            pre = {}
            for i in range(s0, e):
                if i + 4 <= n:
                    pre[i] = (lo.get(int(h[i])), hi.get(int(h[i])))
            for i in range(s0, e):
                if i + 4 <= n:
                    hh = int(h[i])
                    prev[i] = lo.get(hh)
                    lo[hh] = i
                    hi[hh] = pre[i][0]
            ring_lo = max(0, (s0 // W + 1 - p.link_steps) * W)
            for i in range(s0, e, 1):
                if (p.stride2 and i % 2) or i + 4 > n or i % 5:
                    continue
                cands, c, k = [], prev[i], 0
                while k < p.near_depth and c is not None:
                    cands.append(c)
                    k += 1
                    if c < ring_lo:
                        break
                    c = prev[c]
                cands += [x for x in pre[i] if x is not None]
                maxlen = min(n - i, 258, (i // R + 1) * R - i, p.cap)
                best, bd = 0, 0
                for c in cands:
                    if i - c > 32768:
                        continue
                    l = 0
                    while l < maxlen and strip[i + l] == strip[c + l]:
                        l += 1
                    r = min(l, p.rank_bytes)
                    if r > best or (r == best and r and i - c < bd):
                        best, bd = r, i - c
                if best == p.rank_bytes:
                    l, c = 0, i - bd
                    while l < maxlen and strip[i + l] == strip[c + l]:
                        l += 1
                    best = l
                if best == 4 and bd > p.far4_dist:
                    best = 0
                if best >= 4:
                    assert (int(ln[i]), int(ds[i])) == (best, bd), (kw, i)
                else:
                    assert ln[i] == 0, (kw, i)
                checked += 1
        assert checked > 5000
    whole = np.frombuffer(starfleet, np.uint8)
    base = O.compress(whole, O.default_params()).size
    for kw in (dict(near_depth=1, link_steps=1), dict(near_depth=1, link_steps=1, stride2=0, step=512)):
        s = O.compress(whole, O.default_params(recent=1, **kw))
        _check(s, whole)
        assert s.size < base


def test_short_probe_behind_a_skipped_block():
    """Stored fast path, round 5: the block behind a block that took the fast path is searched on its first SFO_SKIP_PROBE
    positions only.  High-entropy data stays stored; a strip that turns compressible again is still coded (its first
    block after the noise loses the matches of the unsearched part of the probe span, nothing else), and everything
    round-trips."""
    rng = np.random.default_rng(8)
    rnd = rng.integers(0, 256, 6 * CHUNK, dtype=np.uint8)
    text = synth.gen_text(4 * CHUNK, seed=9)
    for bb in (0, 4 * CHUNK, 16 * CHUNK):
        s = O.compress(rnd, O.default_params(strip_bytes=bb))
        assert s.size == rnd.size + 5 * 6
        _check(s, rnd)
        mix = np.concatenate([rnd[: 3 * CHUNK], text, rnd[3 * CHUNK:]])
        on, off = O.compress(mix, O.default_params(strip_bytes=bb)), O.compress(mix, O.default_params(strip_bytes=bb, fast_skip=0))
        _check(on, mix)
        assert on.size <= off.size + 3000  # the fast path costs a probe span's worth of matches at most
    # the tokens of a block behind a skipped one: matches only inside the first SFO_SKIP_PROBE positions of its probe span
    two = np.concatenate([rnd[:CHUNK], np.tile(text[:4096], 8)])
    toks, nt = O.strip_tokens(two, O.default_params(strip_bytes=2 * CHUNK))
    R = 512
    for r in range(CHUNK // R, 2 * CHUNK // R):
        t = toks[r * R: r * R + nt[r]]
        if 2048 <= r * R - CHUNK < 8192:
            assert not np.any(t & 0x80000000), r  # unsearched part of the probe span: literals
    assert np.any(toks[(CHUNK + 8192) // R * R:] & 0x80000000)  # the rest of the block is searched again


def test_stored_by_the_probe():
    """Round 6 (probe_span_is_noise): a full block that takes the stored fast path and whose first SFO_SKIP_SPAN bytes are as
    good as uniform has NO tokens, and the plan stores a block with bytes but no tokens; the rest of such a block is never
    looked at.  Not with a forced block type, not for a strip's short last block, not for bytes a Huffman code shortens."""
    rng = np.random.default_rng(12)
    rnd = rng.integers(0, 256, 4 * CHUNK, dtype=np.uint8)
    six = rng.integers(0, 64, 4 * CHUNK, dtype=np.uint8)
    R = 512
    par = O.default_params(strip_bytes=4 * CHUNK)
    toks, nt = O.strip_tokens(rnd, par)
    assert not nt.any()
    s = O.compress(rnd, par)
    assert s.size == rnd.size + 5 * 4
    _check(s, rnd)
    _, nt = O.strip_tokens(rnd, O.default_params(strip_bytes=4 * CHUNK, strategy=3))
    assert nt.sum() == rnd.size  # a forced block type: every byte a literal
    _check(O.compress(rnd, O.default_params(strategy=3)), rnd)
    _, nt = O.strip_tokens(six, par)
    assert nt.sum() == six.size  # six bits per byte: the fast path, but literals under a Huffman code
    assert O.compress(six, par).size < 0.8 * six.size
    # a noise head stores the whole block; the block behind it is probed, and coded
    mix = np.concatenate([rnd[:8192], six[: CHUNK - 8192], six[:CHUNK], rnd[: CHUNK + 8200]])
    _, nt = O.strip_tokens(mix, par)
    per = [int(nt[c // R: (min(c + CHUNK, mix.size) + R - 1) // R].sum()) for c in range(0, mix.size, CHUNK)]
    assert per == [0, CHUNK, 0, 8200], per
    s = O.compress(mix, par)
    _check(s, mix)
    assert s.size < CHUNK + 5 + 0.8 * CHUNK + CHUNK + 5 + 8200 + 5 + 64
    # the rule looks at the probe span only, through the plan's fixed-point entropy
    pw = np.ones(256)
    pw[:64] = 0.3
    skew = rng.choice(256, size=2 * CHUNK, p=pw / pw.sum()).astype(np.uint8)
    _, nt = O.strip_tokens(skew, par)
    assert nt.sum() == skew.size
    # ... and sfo_parse_chunk (a chunk as a strip of its own) decides alike
    ln, ds = O.match_chunk(rnd[:CHUNK], O.default_params(strip_bytes=CHUNK))
    _, nt1 = O.parse_chunk(rnd[:CHUNK], O.default_params(strip_bytes=CHUNK), ln, ds)
    assert not nt1.any()
    # the plan: bytes but no tokens -> stored
    ll = np.zeros(286, np.uint32)
    ll[256] = 1
    pl = O.plan_chunk(ll, np.zeros(30, np.uint32), CHUNK, False, O.default_params())
    assert pl.btype == 0 and pl.out_bytes == CHUNK + 5


def test_plan_stores_all_but_incompressible_chunks_without_a_code():
    """sfo_plan_chunk, round 5: when the fixed block is no shorter than the stored one and the integer entropy estimate of
    the dynamic block comes within SFO_STORE_MARGIN (64) bytes of it, the chunk is stored and no code is built.  The rule
    may give up at most that margin against the exact choice, fires on high-entropy chunks, and leaves every chunk that
    compresses alone."""
    rng = np.random.default_rng(12)
    p, p_dyn = O.default_params(), O.default_params(strategy=3)
    fired = 0
    for kind in range(40):
        if kind < 10:
            data = rng.integers(0, 256, CHUNK, dtype=np.uint8)                       # noise
        elif kind < 20:
            data = rng.integers(0, 256 - 4 * (kind - 9), CHUNK, dtype=np.uint8)      # a slightly smaller alphabet: near the margin
        elif kind < 30:
            data = rng.integers(0, 64, CHUNK, dtype=np.uint8)                        # six bits per byte: compresses, no matches
        else:
            data = synth.gen_text(CHUNK, seed=kind)
        ll = np.bincount(data, minlength=286).astype(np.uint32)
        ll[256] = 1
        d = np.zeros(30, np.uint32)
        auto, dyn = O.plan_chunk(ll, d, CHUNK, False, p), O.plan_chunk(ll, d, CHUNK, False, p_dyn)
        if auto.btype == 0:
            assert auto.out_bytes == CHUNK + 5 and not any(auto.ll_lens)
            assert dyn.out_bytes + 64 >= auto.out_bytes, kind   # the margin is all the rule can cost
            fired += kind < 20
        else:
            assert auto.btype == 2 and auto.out_bytes == dyn.out_bytes and auto.out_bytes < CHUNK + 5, kind
        if kind < 10:
            assert auto.btype == 0, kind
        if kind >= 20:
            assert auto.btype == 2, kind
    assert fired >= 10
    whole = rng.integers(0, 256, 3 * CHUNK + 100, dtype=np.uint8)
    s = O.compress(whole)
    assert s.size == whole.size + 5 * 4
    _check(s, whole)
