"""ctypes binding of the CPU oracle (oracle/libsf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under starflate_amd/ may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class Params(C.Structure):
    _fields_ = [
        ("chunk_bytes", C.c_uint32),
        ("step", C.c_uint32),
        ("hash_bits", C.c_uint32),
        ("region_bytes", C.c_uint32),
        ("min_match", C.c_uint32),
        ("lazy", C.c_uint32),
        ("final_stream", C.c_uint32),
        ("strategy", C.c_uint32),
        ("depth", C.c_uint32),
        ("use_near", C.c_uint32),
        ("long_hash_bytes", C.c_uint32),
        ("chain_depth", C.c_uint32),
        ("cap", C.c_uint32),
        ("fast_skip", C.c_uint32),
        ("far4_dist", C.c_uint32),
        ("container", C.c_uint32),
        ("strip_bytes", C.c_uint32),
        ("x_long_levels", C.c_uint32),
        ("x_long_near", C.c_uint32),
        ("rank_bytes", C.c_uint32),
        ("x_window", C.c_uint32),
        ("x_stride2", C.c_uint32),
        ("use_prev", C.c_uint32),
        ("stride2", C.c_uint32),
        ("run_dist1", C.c_uint32),
        ("recent", C.c_uint32),
        ("near_depth", C.c_uint32),
        ("link_steps", C.c_uint32),
    ]


class Plan(C.Structure):
    _fields_ = [
        ("btype", C.c_uint32),
        ("out_bytes", C.c_uint32),
        ("header_bits", C.c_uint32),
        ("body_bits", C.c_uint32),
        ("ll_lens", C.c_uint8 * 288),
        ("d_lens", C.c_uint8 * 32),
        ("header", C.c_uint8 * 600),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libsf_oracle.so")
        src = os.path.join(ORACLE_DIR, "sf_oracle.c")
        if not os.path.exists(path) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)
        ):
            build()
        L = C.CDLL(path)
        u8p = C.POINTER(C.c_uint8)
        L.sfo_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.sfo_decompress.restype = C.c_int
        L.sfo_read_header.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sfo_read_header.restype = C.c_int
        L.sfo_copy_from_before.argtypes = [C.c_uint16, C.c_void_p, C.c_uint16]
        L.sfo_copy_from_before.restype = None
        L.sfo_canonical_codes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.sfo_canonical_codes.restype = None
        L.sfo_huffman_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.sfo_huffman_decode.restype = C.c_size_t
        L.sfo_default_params.argtypes = [C.POINTER(Params)]
        L.sfo_default_params.restype = None
        L.sfo_compress_bound.argtypes = [C.c_size_t, C.POINTER(Params)]
        L.sfo_compress_bound.restype = C.c_size_t
        L.sfo_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Params)]
        L.sfo_compress.restype = C.c_int
        L.sfo_compress_indexed.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                           C.POINTER(Params), C.c_void_p, C.c_void_p]
        L.sfo_compress_indexed.restype = C.c_int
        L.sfo_crc32.argtypes = [C.c_void_p, C.c_size_t]
        L.sfo_crc32.restype = C.c_uint32
        L.sfo_adler32.argtypes = [C.c_void_p, C.c_size_t]
        L.sfo_adler32.restype = C.c_uint32
        L.sfo_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.sfo_crc32_combine.restype = C.c_uint32
        L.sfo_adler32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.sfo_adler32_combine.restype = C.c_uint32
        L.sfo_strip_tokens.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Params), C.c_void_p, C.c_void_p]
        L.sfo_strip_tokens.restype = C.c_int
        L.sfo_resolve_strip_bytes.argtypes = [C.POINTER(Params), C.c_size_t]
        L.sfo_resolve_strip_bytes.restype = C.c_size_t
        L.sfo_match_chunk.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Params), C.c_void_p, C.c_void_p]
        L.sfo_match_chunk.restype = None
        L.sfo_parse_chunk.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sfo_parse_chunk.restype = None
        L.sfo_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.sfo_histogram.restype = None
        L.sfo_build_lengths.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.sfo_build_lengths.restype = None
        L.sfo_plan_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(Params), C.POINTER(Plan)]
        L.sfo_plan_chunk.restype = None
        _ = u8p
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


def default_params(**kw):
    p = Params()
    lib().sfo_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def decompress(src, dst_cap):
    """-> (status, bytes written, output ndarray[dst_cap])"""
    s = _u8(src)
    dst = np.zeros(max(dst_cap, 1), dtype=np.uint8)
    w = C.c_size_t(0)
    st = lib().sfo_decompress(_ptr(s) if s.size else None, s.size, _ptr(dst), dst_cap, C.byref(w))
    return st, w.value, dst[:dst_cap]


def compress(data, params=None):
    p = params or default_params()
    s = _u8(data)
    cap = lib().sfo_compress_bound(s.size, C.byref(p))
    dst = np.zeros(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().sfo_compress(_ptr(s) if s.size else None, s.size, _ptr(dst), cap, C.byref(n), C.byref(p))
    if rc:
        raise RuntimeError(f"sfo_compress rc={rc}")
    return dst[: n.value].copy()


def compress_indexed(data, params=None):
    """-> (stream, index uint64[nchunks+1], subindex uint32[nchunks, 32, 2])"""
    p = params or default_params()
    s = _u8(data)
    cap = lib().sfo_compress_bound(s.size, C.byref(p))
    dst = np.zeros(cap, dtype=np.uint8)
    nch = max(1, (s.size + p.chunk_bytes - 1) // p.chunk_bytes)
    index = np.zeros(nch + 1, dtype=np.uint64)
    sub = np.zeros((nch, 32, 2), dtype=np.uint32)
    n = C.c_size_t(0)
    rc = lib().sfo_compress_indexed(_ptr(s) if s.size else None, s.size, _ptr(dst), cap, C.byref(n), C.byref(p),
                                    _ptr(index), _ptr(sub))
    if rc:
        raise RuntimeError(f"sfo_compress_indexed rc={rc}")
    return dst[: n.value].copy(), index, sub


def crc32(data):
    s = _u8(data)
    return lib().sfo_crc32(_ptr(s) if s.size else None, s.size)


def adler32(data):
    s = _u8(data)
    return lib().sfo_adler32(_ptr(s) if s.size else None, s.size)


def resolve_strip_bytes(params, n):
    return lib().sfo_resolve_strip_bytes(C.byref(params), n)


def strip_tokens(data, params):
    """Stages n1 + parse over one strip -> (tokens uint32[nregions * R] with region r's tokens at [r * R, r * R +
    ntok[r]), ntok uint32[nregions]); regions count from the strip's start, a chunk's regions are those it covers."""
    s = _u8(data)
    R = params.region_bytes
    nreg = (s.size + R - 1) // R
    tokens = np.zeros((nreg + 1) * R, dtype=np.uint32)
    ntok = np.zeros(nreg + 1, dtype=np.uint32)
    rc = lib().sfo_strip_tokens(_ptr(s) if s.size else None, s.size, C.byref(params), _ptr(tokens), _ptr(ntok))
    if rc:
        raise RuntimeError(f"sfo_strip_tokens rc={rc}")
    return tokens[: nreg * R], ntok[:nreg]


def chunk_tokens(data, params):
    """Per chunk of the whole input (strips of sfo_resolve_strip_bytes): (flat token array, tokens per region)."""
    s = _u8(data)
    sb = resolve_strip_bytes(params, s.size)
    cb, R = params.chunk_bytes, params.region_bytes
    out = []
    for s0 in range(0, max(s.size, 1), sb):
        strip = s[s0:s0 + sb]
        t, nt = strip_tokens(strip, params)
        for c0 in range(0, max(strip.size, 1), cb):
            r0, r1 = c0 // R, (min(c0 + cb, strip.size) + R - 1) // R
            flat = [t[r * R: r * R + nt[r]] for r in range(r0, r1)]
            out.append((np.concatenate(flat) if flat else np.zeros(0, np.uint32), nt[r0:r1].copy(), t[r0 * R: r1 * R].copy()))
    return out


def match_chunk(data, params):
    s = _u8(data)
    ln = np.zeros(s.size, dtype=np.uint16)
    ds = np.zeros(s.size, dtype=np.uint16)
    lib().sfo_match_chunk(_ptr(s), s.size, C.byref(params), _ptr(ln), _ptr(ds))
    return ln, ds


def parse_chunk(data, params, ln, ds):
    s = _u8(data)
    R = params.region_bytes
    nreg = (s.size + R - 1) // R
    tokens = np.zeros(nreg * R, dtype=np.uint32)
    ntok = np.zeros(max(nreg, 1), dtype=np.uint32)
    lib().sfo_parse_chunk(_ptr(s), s.size, C.byref(params), _ptr(ln), _ptr(ds), _ptr(tokens), _ptr(ntok))
    return tokens, ntok[:nreg]


def histogram(tokens, ntok, region_bytes):
    ll = np.zeros(286, dtype=np.uint32)
    d = np.zeros(30, dtype=np.uint32)
    lib().sfo_histogram(_ptr(tokens), _ptr(ntok), ntok.size, region_bytes, _ptr(ll), _ptr(d))
    return ll, d


def build_lengths(freq, maxbits):
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    lens = np.zeros(f.size, dtype=np.uint8)
    lib().sfo_build_lengths(_ptr(f), f.size, maxbits, _ptr(lens))
    return lens


def plan_chunk(ll, d, n_raw, is_last, params):
    plan = Plan()
    ll = np.ascontiguousarray(ll, dtype=np.uint32)
    d = np.ascontiguousarray(d, dtype=np.uint32)
    lib().sfo_plan_chunk(_ptr(ll), _ptr(d), n_raw, int(is_last), C.byref(params), C.byref(plan))
    return plan


def canonical_codes(bitsize):
    b = np.ascontiguousarray(bitsize, dtype=np.uint8)
    out = np.zeros(b.size, dtype=np.uint32)
    lib().sfo_canonical_codes(_ptr(b), b.size, _ptr(out))
    return out


def huffman_decode(bitsize, src, nbits, cap=4096):
    b = np.ascontiguousarray(bitsize, dtype=np.uint8)
    s = _u8(src)
    out = np.zeros(cap, dtype=np.uint16)
    n = lib().sfo_huffman_decode(_ptr(b), b.size, _ptr(s), nbits, _ptr(out), cap)
    return out[:n]
