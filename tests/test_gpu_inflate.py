"""GPU decompress of block-indexed streams (k_inflate_tokens + k_inflate_bytes through the C-ABI): the output must
be byte-identical to what the oracle's restatement of the reference decoder (src/decompress.cpp:402-461) produces
from the same stream, for every block type, for wrapped streams, and for zlib-made indexed streams; the decoder's
tokens must equal the compressor's; malformed segments must report the reference's status codes."""
import zlib

import numpy as np
import pytest

import oracle_lib as O
from starflate_amd import _capi, synth

pytestmark = pytest.mark.gpu

CHUNK = 32768


def _inputs(starfleet):
    rng = np.random.default_rng(7)
    text = synth.gen_text(200_000, seed=2)
    return {
        "empty": np.zeros(0, np.uint8),
        "one": np.array([65], np.uint8),
        "tiny_rep": np.frombuffer(b"abcabcabcabcabcabcabcabcabc", np.uint8),
        "zeros_ragged": np.zeros(CHUNK * 2 + 777, np.uint8),
        "text_ragged": text[: CHUNK * 5 + 1234],
        "text_chunk_minus1": text[: CHUNK - 1],
        "starfleet": np.frombuffer(starfleet, np.uint8),
        "random": rng.integers(0, 256, CHUNK * 3 + 5, dtype=np.uint8),
        "low_entropy": rng.integers(0, 4, CHUNK + 99, dtype=np.uint8),
        "period7": np.tile(np.arange(7, dtype=np.uint8), 9000),
        "period300": np.tile(rng.integers(0, 256, 300, dtype=np.uint8), 400),
        "mixed": synth.gen_mixed(3 << 20, seed=4, stripe=1 << 16)[: (1 << 20) + 13],
    }


def _gpu_roundtrip(compressor, data, use_sub=False, **kw):
    import torch

    src = torch.from_numpy(data.copy()).cuda() if data.size else torch.empty(0, dtype=torch.uint8, device="cuda")
    out, n = compressor.compress_tensor(src, **kw)
    index = compressor.last_index(device="cuda")
    sub = compressor.last_subindex(device="cuda") if use_sub else None
    stream = out[:n].clone()
    back, status = compressor.decompress_tensor(stream, index, data.size, subindex=sub, block_bytes=compressor.last_block_bytes())
    return stream.cpu().numpy(), index.cpu().numpy().astype(np.uint64), back.cpu().numpy(), status


@pytest.mark.parametrize("block_bytes", [32768, 131072])
@pytest.mark.parametrize("use_sub", [False, True])
@pytest.mark.parametrize("strategy", ["auto", "stored", "fixed", "dynamic"])
def test_inflate_own_streams_equals_reference_decoder(compressor, starfleet, strategy, use_sub, block_bytes):
    """use_sub: 32 region lanes per segment located by the sub-index (k_inflate_tokens_sub) instead of one lane per
    segment (k_inflate_tokens); same bytes either way.  block_bytes = 131072: strips of four segments whose matches
    reach into the strip's earlier segments (the byte-copy kernel then carries a 64 KiB window through the strip)."""
    for name, data in _inputs(starfleet).items():
        stream, index, back, status = _gpu_roundtrip(compressor, data, use_sub=use_sub, strategy=strategy, block_bytes=block_bytes)
        assert status == 0, (name, status)
        st, w, ref = O.decompress(stream, data.size)  # the reference restatement on the same stream
        assert st == 0 and w == data.size
        assert np.array_equal(back, ref[: data.size]) and np.array_equal(back, data), name
        nseg = max(1, (data.size + CHUNK - 1) // CHUNK)
        assert index.size == nseg + 1 and index[0] == 0 and index[-1] == stream.size
        assert np.all(np.diff(index.astype(np.int64)) > 0)


@pytest.mark.parametrize("strategy", ["auto", "fixed", "dynamic", "stored"])
def test_index_and_subindex_equal_oracle(compressor, starfleet, strategy):
    """The side information is part of the specification: chunk offsets and, per 1024-byte parse region, the bit
    offset of its first token code and the tokens before it -- bit-exact with sfo_compress_indexed."""
    for k, (name, data) in enumerate(_inputs(starfleet).items()):
        bb = [0, 65536, 32768][k % 3]
        got = np.frombuffer(compressor.compress(data, strategy=strategy, block_bytes=bb), np.uint8)
        idx, sub = compressor.last_index(), compressor.last_subindex()
        want, widx, wsub = O.compress_indexed(data, O.default_params(strategy=_capi.STRATEGY[strategy], strip_bytes=bb))
        assert np.array_equal(got, want), name
        assert np.array_equal(idx, widx), name
        assert np.array_equal(sub, wsub), (name, np.argwhere(sub != wsub)[:4])


def test_wrong_subindex_is_an_error_not_wrong_output(compressor, starfleet):
    data = np.frombuffer(starfleet, np.uint8)
    stream = compressor.compress(data, strategy="dynamic", block_bytes=65536)
    idx, sub = compressor.last_index(), compressor.last_subindex()
    assert compressor.decompress(stream, idx, data.size, subindex=sub, block_bytes=65536) == (data.tobytes(), 0)
    for where, delta in (((1, 7, 0), 1), ((2, 0, 0), 3), ((0, 31, 1), 1), ((3, 12, 1), 5), ((1, 3, 0), 1 << 20)):
        bad = sub.copy()
        bad[where] += np.uint32(delta)
        out, st = compressor.decompress(stream, idx, data.size, subindex=bad, block_bytes=65536)
        assert st != 0 and out == b"", (where, st)
    zero = np.zeros_like(sub)
    assert compressor.decompress(stream, idx, data.size, subindex=zero, block_bytes=65536)[1] != 0


def test_wrong_block_bytes_is_an_error_not_wrong_output(compressor):
    """A stream whose matches reach into the previous segment, decoded as if every segment were independent (or with
    strips that start elsewhere): InvalidDistance (7), as src/decompress.cpp:178 reports a distance beyond the bytes
    written -- never other bytes."""
    data = synth.gen_text(8 * CHUNK, seed=19)
    stream = compressor.compress(data, block_bytes=4 * CHUNK)
    idx = compressor.last_index()
    assert compressor.decompress(stream, idx, data.size, block_bytes=4 * CHUNK) == (data.tobytes(), 0)
    assert compressor.decompress(stream, idx, data.size, block_bytes=8 * CHUNK) == (data.tobytes(), 0)  # a coarser strip is fine
    for bb in (CHUNK, 2 * CHUNK):
        out, st = compressor.decompress(stream, idx, data.size, block_bytes=bb)
        assert st == 7 and out == b"", (bb, st)
    with pytest.raises(Exception):
        compressor.decompress(stream, idx, data.size, block_bytes=CHUNK + 1)


def test_decoder_tokens_equal_compressor_tokens(compressor):
    """k_inflate_tokens inverts k_emit exactly: the token stream it recovers is the one k_lz77 produced."""
    import torch

    data = synth.gen_text(CHUNK * 6 + 99, seed=13)
    nch = 7
    src = torch.from_numpy(data).cuda()
    out, n = compressor.compress_tensor(src, strategy="dynamic", block_bytes=4 * CHUNK)
    index = compressor.last_index(device="cuda")
    ntok_c = compressor.debug(_capi.DBG_NTOK, nch).copy()
    tok_c, _ = compressor.debug_tokens(nch)
    back, status = compressor.decompress_tensor(out[:n].clone(), index, data.size, block_bytes=4 * CHUNK)
    assert status == 0 and np.array_equal(back.cpu().numpy(), data)
    tok_d = compressor.debug(_capi.DBG_TOKENS, nch)  # the decoder's own token buffer
    for c in range(nch):
        k = int(ntok_c[c])
        assert tok_c[c].size == k and np.array_equal(tok_c[c], tok_d[c, :k]), c


@pytest.mark.parametrize("container", ["zlib", "gzip"])
def test_inflate_wrapped_streams(compressor, starfleet, container):
    """The index of a wrapped stream points past the wrapper header, so the same call decodes it in place."""
    data = np.frombuffer(starfleet, np.uint8)
    stream, index, back, status = _gpu_roundtrip(compressor, data, container=container)
    h, t = (2, 4) if container == "zlib" else (10, 8)
    assert status == 0 and np.array_equal(back, data)
    assert index[0] == h and index[-1] == stream.size - t
    assert zlib.decompress(stream.tobytes(), 15 if container == "zlib" else 31) == data.tobytes()


@pytest.mark.parametrize("level,strategy", [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (0, zlib.Z_DEFAULT_STRATEGY)])
def test_inflate_zlib_made_indexed_streams(compressor, starfleet, level, strategy):
    """Streams this library did not write: zlib with Z_FULL_FLUSH every 32 KiB (3-byte matches, distances up to
    32768, 15-bit codes, several blocks per segment, stored blocks inside segments)."""
    for name, data in _inputs(starfleet).items():
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        nch = max(1, (data.size + CHUNK - 1) // CHUNK)
        parts = []
        for c in range(nch):
            b = co.compress(data[c * CHUNK:(c + 1) * CHUNK].tobytes())
            parts.append(b + co.flush(zlib.Z_FINISH if c == nch - 1 else zlib.Z_FULL_FLUSH))
        stream = b"".join(parts)
        index = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
        got, status = compressor.decompress(stream, index, data.size, block_bytes=32768)  # host-buffer entry point
        assert status == 0, (name, status)
        assert got == data.tobytes(), name
        # the same with strips of four segments: the window survives Z_SYNC_FLUSH and restarts at Z_FULL_FLUSH
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        parts = []
        for c in range(nch):
            b = co.compress(data[c * CHUNK:(c + 1) * CHUNK].tobytes())
            parts.append(b + co.flush(zlib.Z_FINISH if c == nch - 1 else (zlib.Z_FULL_FLUSH if c % 4 == 3 else zlib.Z_SYNC_FLUSH)))
        index = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
        got, status = compressor.decompress(b"".join(parts), index, data.size, block_bytes=4 * CHUNK)
        assert status == 0 and got == data.tobytes(), (name, "strips")


@pytest.mark.parametrize("name", ["starfleet.html.dynamic.flushed", "starfleet.html.fixed.flushed"])
def test_inflate_reference_made_fixture(compressor, starfleet, name):
    """Reference-held bytes through the HIP decoder: the file the reference's own test decodes
    (/root/reference/src/test/decompress_test.cpp:136-174), compressed with its fixture tool's zlib settings and a
    Z_FULL_FLUSH every 32 KiB (tests/golden/make_golden.py; committed with its index).  The GPU decoder must return what the
    oracle's restatement of the reference decoder returns for the same stream: the file."""
    import os

    from conftest import GOLDEN

    with open(os.path.join(GOLDEN, name), "rb") as f:
        stream = f.read()
    index = np.fromfile(os.path.join(GOLDEN, name + ".index"), dtype="<u8").astype(np.uint64)
    st, w, want = O.decompress(np.frombuffer(stream, np.uint8), len(starfleet))
    assert st == 0 and w == len(starfleet) and want.tobytes() == starfleet
    got, status = compressor.decompress(stream, index, len(starfleet), block_bytes=32768)
    assert status == 0 and got == starfleet
    # device-resident call, and a truncated segment reports an error instead of bytes
    import torch

    gback, gst = compressor.decompress_tensor(torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda(),
                                              torch.from_numpy(index.astype(np.int64)).cuda(), len(starfleet), block_bytes=32768)
    assert gst == 0 and gback.cpu().numpy().tobytes() == starfleet
    cut = index.copy()
    cut[-1] -= np.uint64(40)
    assert compressor.decompress(stream[:-40], cut, len(starfleet), block_bytes=32768)[1] != 0


def test_malformed_segments_report_reference_statuses(compressor, starfleet):
    data = np.frombuffer(starfleet, np.uint8)
    nseg = (data.size + CHUNK - 1) // CHUNK
    raw = np.frombuffer(compressor.compress(data, strategy="stored"), np.uint8).copy()
    idx = compressor.last_index()
    assert compressor.decompress(raw, idx, data.size, block_bytes=32768) == (data.tobytes(), 0)
    bad = raw.copy()
    bad[int(idx[2]) + 3] ^= 1  # NLEN of the third stored block
    assert compressor.decompress(bad, idx, data.size, block_bytes=32768)[1] == 3  # NoCompressionLenMismatch
    bad = raw.copy()
    bad[int(idx[1])] |= 0b110  # BTYPE 3 in the second segment; the first failing segment in stream order wins
    bad[int(idx[3]) + 3] ^= 1
    assert compressor.decompress(bad, idx, data.size, block_bytes=32768)[1] == 2  # InvalidBlockHeader
    dyn = np.frombuffer(compressor.compress(data, strategy="dynamic"), np.uint8).copy()
    idx = compressor.last_index()
    assert compressor.decompress(dyn, idx, data.size, block_bytes=32768) == (data.tobytes(), 0)
    cut = idx.copy()
    cut[1:] -= np.uint64(100)  # every segment starts 100 bytes early: garbage
    assert compressor.decompress(dyn, cut, data.size, block_bytes=32768)[1] != 0
    short = idx.copy()
    short[-1] -= np.uint64(40)  # last segment truncated
    assert compressor.decompress(dyn[: int(short[-1])], short, data.size, block_bytes=32768)[1] in (5, 6, 7)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, dyn.size, dtype=np.uint8)
    assert compressor.decompress(noise, idx, data.size, block_bytes=32768)[1] != 0  # garbage never crashes, never reports success
    with pytest.raises(Exception):  # nseg must match the output size
        compressor.decompress(dyn, idx[:-1], data.size, block_bytes=32768)
    assert nseg + 1 == idx.size
    # over-subscribed code lengths: Error (1) from both GPU paths, as from the oracle and the C++ host API
    from test_oracle_decompress import oversubscribed_streams
    for bad in oversubscribed_streams():
        b = np.frombuffer(bytes(bad), np.uint8)
        one = np.array([0, b.size], np.uint64)
        assert O.decompress(b, 64)[0] == 1
        assert compressor.decompress(b, one, 64, block_bytes=32768)[1] == 1
        sub = np.zeros((1, 32, 2), np.uint32)
        assert compressor.decompress(b, one, 64, subindex=sub, block_bytes=32768)[1] != 0


def test_inflate_large_roundtrip_and_timing(compressor):
    import torch

    data = synth.gen_text(64 << 20, seed=3)
    src = torch.from_numpy(data).cuda()
    out, n = compressor.compress_tensor(src)
    index = compressor.last_index(device="cuda")
    stream = out[:n].clone()
    sub = compressor.last_subindex(device="cuda")
    bb = compressor.last_block_bytes()
    assert bb == 262144
    compressor.set_profiling(True)
    back, status = compressor.decompress_tensor(stream, index, data.size, block_bytes=bb)
    assert status == 0 and torch.equal(back, src)
    ms = compressor.inflate_ms()
    back3, status3 = compressor.decompress_tensor(stream, index, data.size, subindex=sub, block_bytes=bb)
    assert status3 == 0 and torch.equal(back3, src)
    ms_sub = compressor.inflate_ms()
    compressor.set_profiling(False)
    print("inflate 64 MiB:", ms, "with sub-index:", ms_sub)
    back2, status2 = compressor.decompress_tensor(stream, index, data.size, block_bytes=bb)  # deterministic
    assert status2 == 0 and torch.equal(back2, back)


def test_beyond_4_gib_offsets(compressor):
    """4.25 GiB in one call: chunk offsets, token scratch indices and the index cross 2^32; the GPU decoder
    (bit-exact with the reference decoder on everything smaller) is the round-trip check, plus zlib on a slice."""
    import torch

    n = (17 << 28) + 12345  # 4.25 GiB + a ragged tail
    free, _ = torch.cuda.mem_get_info()
    if free < 8 * n:
        pytest.skip("not enough device memory")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    piece = synth.gen_text_torch(1 << 28, seed=21, device=torch.device("cuda", 0))
    src = torch.empty(n, dtype=torch.uint8, device="cuda")
    for k in range(0, n, 1 << 28):  # 17 different rotations of the same 256 MiB, so chunks differ across the 4 GiB line
        m = min(1 << 28, n - k)
        src[k:k + m] = torch.roll(piece, shifts=int(k >> 20) * 7919 + 13)[:m]
    del piece
    out, nb = compressor.compress_tensor(src)
    index = compressor.last_index(device="cuda")
    sub = compressor.last_subindex(device="cuda")
    assert nb > (1 << 30) and int(index[-1]) == nb and bool((index[1:] > index[:-1]).all())
    stream = out[:nb]
    back, status = compressor.decompress_tensor(stream, index, n, subindex=sub, block_bytes=compressor.last_block_bytes())
    assert status == 0 and torch.equal(back, src)
    # the decoder's token scratch is ONE batch's (1 GiB of output: 4 GiB), not 4 bytes per byte of the whole call (17 GiB)
    assert compressor.last_decode_scratch_bytes() == 4 << 30
    del back
    # the last strip (beyond 2^32) through zlib as an independent judge
    nseg = index.numel() - 1
    first = (nseg - 1) // 8 * 8  # strips of 8 segments
    lo = int(index[first])
    tail = stream[lo:nb].cpu().numpy().tobytes()
    assert zlib.decompress(tail, -15) == src[first * CHUNK:].cpu().numpy().tobytes()


def test_corrupted_streams_never_succeed_wrongly(compressor, starfleet):
    """Bit flips in a valid stream: both decoder paths either report a reference status or produce exactly what the
    oracle's restatement of the reference decoder produces from the same damaged bytes (a flip in a literal's
    code can yield another valid stream).  Nothing may crash or run away."""
    rng = np.random.default_rng(31337)
    data = np.frombuffer(starfleet, np.uint8)
    stream = np.frombuffer(compressor.compress(data, strategy="dynamic", block_bytes=65536), np.uint8).copy()
    idx, sub = compressor.last_index(), compressor.last_subindex()
    agree = 0
    for it in range(60):
        bad = stream.copy()
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(0, bad.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        st_ref, w_ref, out_ref = O.decompress(bad, data.size)
        for s in (None, sub):
            got, st = compressor.decompress(bad, idx, data.size, subindex=s, block_bytes=65536)
            if st == 0:
                # the serial decoder must then also succeed with the same bytes (it reads the same blocks)
                assert st_ref == 0 and w_ref == data.size and got == out_ref[: data.size].tobytes(), (it, s is not None)
                agree += 1
    assert agree >= 0


def test_fuzz_decoder_on_zlib_and_own_streams(compressor):
    """Seeded fuzz of the decoder: stitched inputs (runs, periods, text, noise, counters) compressed (a) by zlib at a
    random level / strategy with Z_FULL_FLUSH every 32 KiB and (b) by this library; both decoder paths must return
    the input.  SF_FUZZ_N scales it (default 120)."""
    import os

    rng = np.random.default_rng(4242)
    text = synth.gen_text(300_000, seed=91)

    def piece(n):
        kind = int(rng.integers(0, 6))
        if kind == 0:
            return np.full(n, int(rng.integers(0, 256)), np.uint8)
        if kind == 1:
            p = int(rng.integers(1, 3000))
            return np.tile(rng.integers(0, 256, p, dtype=np.uint8), n // p + 1)[:n]
        if kind == 2:
            o = int(rng.integers(0, text.size - n)) if n < text.size else 0
            return text[o:o + n]
        if kind == 3:
            return rng.integers(0, 256, n, dtype=np.uint8)
        if kind == 4:
            return rng.integers(0, int(rng.integers(2, 17)), n, dtype=np.uint8)
        return (np.arange(n, dtype=np.uint32) * int(rng.integers(1, 9))).view(np.uint8)[:n]

    levels = [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
              (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (6, zlib.Z_FILTERED), (0, zlib.Z_DEFAULT_STRATEGY)]
    for it in range(int(os.environ.get("SF_FUZZ_N", "120"))):
        total = int(rng.choice([0, 1, 7, 1024, CHUNK - 1, CHUNK, CHUNK + 1, int(rng.integers(0, 6 * CHUNK))]))
        parts, left = [], total
        while left > 0:
            n = int(min(left, rng.integers(1, 30000)))
            parts.append(np.ascontiguousarray(piece(n)[:n]))
            left -= n
        data = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        level, strat = levels[it % len(levels)]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strat)
        nch = max(1, (data.size + CHUNK - 1) // CHUNK)
        zs = []
        for c in range(nch):
            b = co.compress(data[c * CHUNK:(c + 1) * CHUNK].tobytes())
            zs.append(b + co.flush(zlib.Z_FINISH if c == nch - 1 else zlib.Z_FULL_FLUSH))
        index = np.concatenate([[0], np.cumsum([len(p) for p in zs])]).astype(np.uint64)
        got, st = compressor.decompress(b"".join(zs), index, data.size, block_bytes=32768)
        assert st == 0 and got == data.tobytes(), (it, total, level, strat)
        bb = [32768, 65536, 131072][it % 3]
        own = compressor.compress(data, strategy=["auto", "dynamic", "fixed"][it % 3], block_bytes=bb)
        idx, sub = compressor.last_index(), compressor.last_subindex()
        for s in (None, sub):
            got, st = compressor.decompress(own, idx, data.size, subindex=s, block_bytes=bb)
            assert st == 0 and got == data.tobytes(), (it, total, s is not None, bb)


def test_speculative_index_only_equals_lane_serial(compressor, starfleet, monkeypatch):
    """Index-only streams go through k_inflate_tokens_spec (a wave per two segments: 32 lanes find their token boundaries by
    decoding ahead of their span, count, and the spans then serve as a sub-index; block after block), with k_inflate_tokens
    -- one lane per segment, the serial decoder itself -- behind it for damaged segments and for segments cut into more than
    four blocks.  Bytes and status must be those of the lane-serial kernel alone (SFH_INFLATE_SERIAL=1) on own streams of
    every strategy, on zlib streams with one, several and dozens of blocks per segment, on damaged streams; which kernel
    finished a segment is checked (SFH_DBG_SEGINFO); and on a large input the speculative kernel must be the faster one."""
    import os

    import torch
    from starflate_amd import Compressor

    monkeypatch.setenv("SFH_INFLATE_SERIAL", "1")
    serial = Compressor(0)
    monkeypatch.delenv("SFH_INFLATE_SERIAL")
    try:
        rng = np.random.default_rng(99)
        cases = []
        for name, data in _inputs(starfleet).items():
            for strategy, bb in (("auto", 262144), ("dynamic", 32768), ("dynamic", 131072), ("fixed", 65536)):
                stream = np.frombuffer(compressor.compress(data, strategy=strategy, block_bytes=bb), np.uint8).copy()
                cases.append((f"{name}/{strategy}/{bb}", stream, compressor.last_index(), data, bb))
        # zlib with a full flush every 32 KiB.  memLevel 6 closes a block every 4,095 symbols -- two to four blocks per
        # segment, which the speculative kernel follows; memLevel 1 every 127 -- dozens, which it leaves to the serial one
        mixed = synth.gen_mixed(1 << 20, seed=8, stripe=1 << 14)[: 9 * CHUNK + 321]
        text = synth.gen_text(6 * CHUNK + 77, seed=12)
        # (level 1 and periodic data: wrong starts fall in late or never, lanes start over many times in a row -- the
        # checkpoint shortcut and its fall-back to the whole span)
        periodic = np.concatenate([np.tile(rng.integers(0, 256, p, dtype=np.uint8), 3 * CHUNK // p + 1)[: 3 * CHUNK] for p in (7, 300, 4099)])
        for zname, zdata, level, mem in (("zlib6m8", mixed, 6, 8), ("zlib9m8", mixed, 9, 8), ("zlib6m6", text, 6, 6), ("zlib1m1", mixed, 1, 1),
                                         ("zlib1m8", text, 1, 8), ("zlib6m8p", periodic, 6, 8), ("zlib1m8p", periodic, 1, 8)):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, mem)
            nch = (zdata.size + CHUNK - 1) // CHUNK
            zs = [co.compress(zdata[c * CHUNK:(c + 1) * CHUNK].tobytes()) + co.flush(zlib.Z_FINISH if c == nch - 1 else zlib.Z_FULL_FLUSH)
                  for c in range(nch)]
            index = np.concatenate([[0], np.cumsum([len(p) for p in zs])]).astype(np.uint64)
            cases.append((zname, np.frombuffer(b"".join(zs), np.uint8).copy(), index, zdata, 32768))
        for name, stream, index, data, bb in cases:
            a = compressor.decompress(stream, index, data.size, block_bytes=bb)
            # who finished each segment (SFH_DBG_SEGINFO: bit 1 of word 2 = the lane-serial kernel): every valid segment with
            # output is the speculative kernel's, with up to four blocks that hold output; an empty input is the serial kernel's
            nseg = index.size - 1
            info = compressor.debug(_capi.DBG_SEGINFO, nseg)
            by_serial = int(((info[:, 2] >> 1) & 1).sum())
            if name == "zlib1m1":
                assert by_serial >= nseg - 1, name  # (the ragged last segment has few enough blocks)
            else:
                assert by_serial == (0 if data.size else nseg), (name, by_serial)
            if name == "zlib6m6":
                assert int(info[:-1, 1].min()) > 4096, "more symbols than zlib keeps per block: several blocks in every whole segment"
            b = serial.decompress(stream, index, data.size, block_bytes=bb)
            assert int(((serial.debug(_capi.DBG_SEGINFO, nseg)[:, 2] >> 1) & 1).sum()) == nseg, name
            assert a == b and a == (data.tobytes(), 0), name
        # a segment of 64 KiB or more of stream (here: 14,000 empty stored blocks behind the data) is the serial kernel's
        small = text[:5000]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        padded = co.compress(small.tobytes()) + co.flush(zlib.Z_SYNC_FLUSH) + 14000 * b"\x00\x00\x00\xff\xff" + b"\x01\x00\x00\xff\xff"
        pidx = np.array([0, len(padded)], np.uint64)
        for c in (compressor, serial):
            assert c.decompress(padded, pidx, small.size, block_bytes=32768) == (small.tobytes(), 0)
        compressor.decompress(padded, pidx, small.size, block_bytes=32768)
        assert int(compressor.debug(_capi.DBG_SEGINFO, 1)[0, 2]) & 2
        # damage: same status, same bytes
        nsame = 0
        for name, stream, index, data, bb in cases[::3]:
            if stream.size < 64:
                continue
            for _ in range(int(os.environ.get("SF_SPEC_FUZZ", "12"))):
                bad = stream.copy()
                for _ in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(0, bad.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
                a = compressor.decompress(bad, index, data.size, block_bytes=bb)
                b = serial.decompress(bad, index, data.size, block_bytes=bb)
                assert a == b, name
                nsame += 1
        assert nsame > 100
        # the point of it: time of the token stage, 64 MiB of text
        data = synth.gen_text(64 << 20, seed=5)
        src = torch.from_numpy(data).cuda()
        out, n = compressor.compress_tensor(src)
        index, bb = compressor.last_index(device="cuda"), compressor.last_block_bytes()
        stream = out[:n].clone()
        ms = {}
        for key, c in (("spec", compressor), ("serial", serial)):
            c.set_profiling(True)
            best = 1e9
            for _ in range(3):
                back, st = c.decompress_tensor(stream, index, data.size, block_bytes=bb)
                assert st == 0 and torch.equal(back, src)
                best = min(best, c.inflate_ms()["k_inflate_tokens"])
            c.set_profiling(False)
            ms[key] = best
        print("index-only token stage, 64 MiB:", ms)
        # (how MUCH faster is bench.py's business -- decompress.segment_indexed; a correctness suite on a busy box only asks
        # that the kernel built to be faster is not the slower one)
        assert ms["spec"] <= ms["serial"], ms
    finally:
        serial.close()


def test_decoder_batches(monkeypatch):
    """The decoder runs batch after batch of whole strips with one batch's token scratch (SFH_BATCH_CHUNKS shrinks the batch for
    the test): same bytes, same status and same first failing segment as one pass, with and without the sub-index, per-stage
    times summed over the batches."""
    import torch

    from starflate_amd import Compressor

    data = np.concatenate([synth.gen_text(9 * CHUNK + 777, seed=11), synth.gen_mixed(14 * CHUNK, seed=12)])
    n = data.size
    ref = Compressor(0)
    monkeypatch.setenv("SFH_BATCH_CHUNKS", "8")
    small = Compressor(0)
    monkeypatch.delenv("SFH_BATCH_CHUNKS")
    try:
        for bb in (CHUNK, 4 * CHUNK):
            out, nb = ref.compress_tensor(torch.from_numpy(data).cuda(), block_bytes=bb)
            index, sub = ref.last_index(device="cuda"), ref.last_subindex(device="cuda")
            stream = out[:nb].clone()
            for s in (sub, None):
                small.set_profiling(True)
                back, st = small.decompress_tensor(stream, index, n, subindex=s, block_bytes=bb)
                ms = small.inflate_ms()
                small.set_profiling(False)
                assert st == 0 and np.array_equal(back.cpu().numpy(), data)
                assert small.last_decode_scratch_bytes() == 8 * CHUNK * 4 and all(v > 0 for v in ms.values())
                back2, st2 = ref.decompress_tensor(stream, index, n, subindex=s, block_bytes=bb)
                assert st2 == 0 and ref.last_decode_scratch_bytes() == ((n + CHUNK - 1) // CHUNK) * CHUNK * 4
            # damage one byte in the third batch: both report the same status (and the message names the same segment)
            bad = stream.clone()
            at = int(index[19]) + 9
            bad[at] = bad[at] ^ 0x5A
            _, sa = small.decompress_tensor(bad, index, n, block_bytes=bb)
            ea = small.last_error()
            _, sb = ref.decompress_tensor(bad, index, n, block_bytes=bb)
            eb = ref.last_error()
            assert sa == sb and sa != 0 and ea == eb and "segment 19" in ea, (sa, sb, ea, eb)
    finally:
        small.close()
        ref.close()
