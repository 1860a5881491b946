"""The per-segment DEFLATE decode that every GPU lane runs (starflate_amd/csrc/sf_inflate_core.h), compiled
for the host and checked on CPU: its tokens, expanded, must equal what the oracle's restatement of the
reference decoder (sfo_decompress, src/decompress.cpp:402-461) produces for the same segment, on streams made
by the encoder specification (all block types) and on zlib-made block-indexed streams (Z_FULL_FLUSH every
32 KiB); malformed segments must report the reference's status codes.  (The GPU kernels themselves are
covered by tests/test_gpu_inflate.py.)"""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

import oracle_lib as O
from conftest import ROOT
from starflate_amd import synth

CHUNK = 32768
CLANG = "/opt/rocm/llvm/bin/clang++"


@pytest.fixture(scope="module")
def core(tmp_path_factory):
    so = tmp_path_factory.mktemp("sfi") / "libsfi.so"
    subprocess.check_call([CLANG, "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "cpp", "inflate_core_host.cpp"), "-o", str(so)])
    L = C.CDLL(str(so))
    L.sfi_decode_segment.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p,
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.sfi_decode_segment.restype = C.c_uint32
    L.sfi_decode_segment_sub.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.sfi_decode_segment_sub.restype = C.c_uint32
    L.sfi_expand_tokens.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.sfi_expand_tokens.restype = C.c_longlong
    return L


def decode_segment(L, stream, lo, hi, out_n):
    """-> (status, bytes or None)"""
    buf = np.zeros(stream.size + 3, np.uint8)  # the decoder reads whole dwords; src_n bounds them
    buf[: stream.size] = stream
    tok = np.zeros(CHUNK + 4, np.uint32)
    ntok, raw, raw_off = C.c_uint32(), C.c_uint32(), C.c_uint64()
    st = L.sfi_decode_segment(buf.ctypes.data, stream.size, lo, hi, out_n, tok.ctypes.data, C.byref(ntok),
                              C.byref(raw), C.byref(raw_off))
    if st:
        return st, None
    if raw.value:
        assert ntok.value == 0
        return 0, stream[raw_off.value: raw_off.value + out_n].copy()
    out = np.zeros(max(out_n, 1), np.uint8)
    n = L.sfi_expand_tokens(tok.ctypes.data, ntok.value, out.ctypes.data, out_n)
    assert n == out_n, (n, out_n)
    return 0, out[:out_n]


def indexed_by_spec(data, **kw):
    """stream + index made by the encoder specification, one segment per 32 KiB chunk"""
    nch = max(1, (data.size + CHUNK - 1) // CHUNK)
    parts = [O.compress(data[c * CHUNK:(c + 1) * CHUNK], O.default_params(final_stream=int(c == nch - 1), **kw)) for c in range(nch)]
    idx = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.uint64)
    return np.concatenate(parts), idx


def indexed_by_zlib(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """zlib's own block-indexed stream: Z_FULL_FLUSH after every 32 KiB (byte-aligned, window reset)"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    parts = []
    nch = max(1, (data.size + CHUNK - 1) // CHUNK)
    for c in range(nch):
        b = co.compress(data[c * CHUNK:(c + 1) * CHUNK].tobytes())
        b += co.flush(zlib.Z_FINISH if c == nch - 1 else zlib.Z_FULL_FLUSH)
        parts.append(np.frombuffer(b, np.uint8))
    idx = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.uint64)
    return np.concatenate(parts), idx


def check_all_segments(L, stream, idx, data):
    st, w, whole = O.decompress(stream, data.size)  # the reference restatement on the whole stream
    oracle_ok = st == 0 and w == data.size and np.array_equal(whole, data)
    for c in range(idx.size - 1):
        out_n = min(CHUNK, data.size - c * CHUNK)
        s, got = decode_segment(L, stream, int(idx[c]), int(idx[c + 1]), out_n)
        assert s == 0, (c, s)
        assert np.array_equal(got, data[c * CHUNK: c * CHUNK + out_n]), c
    return oracle_ok


def _inputs(starfleet):
    rng = np.random.default_rng(5)
    text = synth.gen_text(5 * CHUNK + 1234, seed=2)
    return {
        "empty": np.zeros(0, np.uint8), "one": np.array([7], np.uint8), "text": text, "text_m1": text[: CHUNK - 1],
        "html": np.frombuffer(starfleet, np.uint8), "zeros": np.zeros(2 * CHUNK + 5, np.uint8),
        "random": rng.integers(0, 256, CHUNK * 2 + 17, dtype=np.uint8), "period7": np.tile(np.arange(7, dtype=np.uint8), 9000),
        "low_entropy": rng.integers(0, 3, CHUNK + 99, dtype=np.uint8), "mixed": synth.gen_mixed(300_000, seed=4, stripe=1 << 15),
    }


@pytest.mark.parametrize("strategy", [0, 1, 2, 3])
def test_spec_streams_every_block_type(core, starfleet, strategy):
    for name, data in _inputs(starfleet).items():
        stream, idx = indexed_by_spec(data, strategy=strategy)
        assert check_all_segments(core, stream, idx, data), name  # and the reference restatement agrees


@pytest.mark.parametrize("level,strategy", [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (0, zlib.Z_DEFAULT_STRATEGY)])
def test_zlib_made_indexed_streams(core, starfleet, level, strategy):
    """zlib's streams use everything the format allows (3-byte matches, 15-bit codes, code-length runs that cross
    from the literal/length into the distance lengths -- where the reference decoder itself goes wrong, SURVEY.md
    8(c) hazard A): the lane decoder follows RFC 1951 and must reproduce the input on all of them."""
    agree = 0
    for name, data in _inputs(starfleet).items():
        stream, idx = indexed_by_zlib(data, level, strategy)
        assert zlib.decompress(stream.tobytes(), -15) == data.tobytes()
        agree += check_all_segments(core, stream, idx, data)
    assert agree >= 1  # the reference restatement decodes at least the trivial ones identically


def test_malformed_segments_report_reference_statuses(core, starfleet):
    data = np.frombuffer(starfleet, np.uint8)[:CHUNK]
    L = core
    # stored: rose / bud vectors of src/test/decompress_test.cpp:62-110
    s = np.array([0b000, 4, 0, 0xFB, 0xFF] + list(b"rose") + [0b001, 3, 0, 0xFC, 0xFF] + list(b"bud"), np.uint8)
    st, got = decode_segment(L, s, 0, s.size, 7)
    assert st == 0 and bytes(got) == b"rosebud"
    assert decode_segment(L, s, 0, s.size, 6)[0] == 4          # DstTooSmall
    assert decode_segment(L, s, 0, 5, 7)[0] == 5               # SrcTooSmall (payload cut)
    bad = s.copy(); bad[3] ^= 1
    assert decode_segment(L, bad, 0, s.size, 7)[0] == 3        # NoCompressionLenMismatch
    assert decode_segment(L, np.array([0b111], np.uint8), 0, 1, 0)[0] == 2   # BTYPE 3
    assert decode_segment(L, np.zeros(0, np.uint8), 0, 0, 0)[0] == 2         # empty input: InvalidBlockHeader
    # fixed block: literal 'a' then a match reaching before the start -> InvalidDistance
    def fixed_bits(codes):
        bits = [1, 1, 0]  # BFINAL, BTYPE=01 (LSB first)
        for val, n, msb in codes:
            bits += [(val >> (n - 1 - k)) & 1 for k in range(n)] if msb else [(val >> k) & 1 for k in range(n)]
        bits += [0] * (-len(bits) % 8)
        return np.packbits(np.array(bits, np.uint8), bitorder="little")
    lit_a = (0x30 + ord("a"), 8, True)
    len3 = (0b0000001, 7, True)   # symbol 257
    dist2 = (1, 5, True)          # distance code 1 = distance 2
    eob = (0, 7, True)
    assert decode_segment(L, fixed_bits([lit_a, len3, dist2, eob]), 0, 4, 4)[0] == 7
    dist1 = (0, 5, True)
    st, got = decode_segment(L, fixed_bits([lit_a, len3, dist1, eob]), 0, 4, 4)
    assert st == 0 and bytes(got) == b"aaaa"
    assert decode_segment(L, fixed_bits([lit_a, (0b11000110, 8, True), dist1, eob]), 0, 4, 300)[0] == 6  # symbol 286
    assert decode_segment(L, fixed_bits([lit_a, len3, (30, 5, True), eob]), 0, 4, 4)[0] == 7              # distance code 30
    # truncated dynamic stream, garbage, wrong promised size
    stream, idx = indexed_by_spec(data, strategy=3)
    assert decode_segment(L, stream, 0, int(idx[1]) // 2, CHUNK)[0] in (5, 6, 7, 1)
    assert decode_segment(L, stream, 0, int(idx[1]), CHUNK - 1)[0] == 4
    assert decode_segment(L, stream, 0, int(idx[1]), CHUNK)[0] == 0
    # over-subscribed code lengths: Error, as from the oracle and the C++ host API (one status from every decoder)
    from test_oracle_decompress import oversubscribed_streams
    for bad in oversubscribed_streams():
        b = np.frombuffer(bytes(bad), np.uint8)
        assert decode_segment(L, b, 0, b.size, 64)[0] == 1
    rng = np.random.default_rng(1)
    for _ in range(300):  # random garbage never crashes and never reports success with the wrong size
        g = rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8)
        st, got = decode_segment(L, g, 0, g.size, int(rng.integers(0, 2000)))
        assert st in range(8)


@pytest.mark.parametrize("strategy", [0, 1, 2, 3])
def test_sub_indexed_regions(core, starfleet, strategy):
    """The accelerated path for this library's own streams: 32 region lanes per segment, each starting at the bit
    offset / token index the sub-index names (oracle: sfo_compress_indexed).  Region by region the tokens must be
    the ones the generic whole-segment decode finds, and expand to the input."""
    L = core
    for name, data in _inputs(starfleet).items():
        stream, idx, sub = O.compress_indexed(data, O.default_params(strategy=strategy))
        assert np.array_equal(stream, O.compress(data, O.default_params(strategy=strategy)))
        buf = np.zeros(stream.size + 3, np.uint8)
        buf[: stream.size] = stream
        for c in range(idx.size - 1):
            out_n = min(CHUNK, data.size - c * CHUNK)
            tok = np.zeros(CHUNK + 4, np.uint32)
            ntok, raw, raw_off = C.c_uint32(), C.c_uint32(), C.c_uint64()
            sc = np.ascontiguousarray(sub[c])
            st = L.sfi_decode_segment_sub(buf.ctypes.data, stream.size, int(idx[c]), int(idx[c + 1]), out_n, sc.ctypes.data,
                                          tok.ctypes.data, C.byref(ntok), C.byref(raw), C.byref(raw_off))
            assert st == 0, (name, c, st)
            want = data[c * CHUNK: c * CHUNK + out_n]
            if raw.value:
                assert np.array_equal(stream[raw_off.value: raw_off.value + out_n], want) and not sc.any()
                continue
            tok2 = np.zeros(CHUNK + 4, np.uint32)
            n2, r2, o2 = C.c_uint32(), C.c_uint32(), C.c_uint64()
            assert L.sfi_decode_segment(buf.ctypes.data, stream.size, int(idx[c]), int(idx[c + 1]), out_n, tok2.ctypes.data,
                                        C.byref(n2), C.byref(r2), C.byref(o2)) == 0
            assert ntok.value == n2.value and np.array_equal(tok[: n2.value], tok2[: n2.value]), (name, c)
            out = np.zeros(max(out_n, 1), np.uint8)
            assert L.sfi_expand_tokens(tok.ctypes.data, ntok.value, out.ctypes.data, out_n) == out_n
            assert np.array_equal(out[:out_n], want), (name, c)
            bad = sc.copy()
            bad[5, 0] += 1  # a sub-index that disagrees with the stream is an error, never wrong output
            st = L.sfi_decode_segment_sub(buf.ctypes.data, stream.size, int(idx[c]), int(idx[c + 1]), out_n, bad.ctypes.data,
                                          tok.ctypes.data, C.byref(ntok), C.byref(raw), C.byref(raw_off))
            assert st != 0 or out_n <= 4 * 1024
