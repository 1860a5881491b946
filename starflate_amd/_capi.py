"""ctypes binding of include/starflate_hip.h (libstarflate_hip.so).

There is no CPU fallback: if the library is missing or no HIP device is present,
every compute entry point raises.
"""
import ctypes as C
import os

from . import build as _build

NSTAGES = 5
INFLATE_NSTAGES = 2
SEGMENT_BYTES = 32768
STRATEGY = {"auto": 0, "stored": 1, "fixed": 2, "dynamic": 3}
CONTAINER = {"raw": 0, "zlib": 1, "gzip": 2}
DBG_NTOK, DBG_TOKENS, DBG_HIST, DBG_PLAN, DBG_LENS, DBG_OFFSETS, DBG_STAMPS = range(7)
DBG_SUBINDEX = 7
DBG_ITEMS, DBG_NITEMS = 8, 9
DBG_SEGINFO = 10
DEFAULT_BLOCK_BYTES = 262144
LARGE_BLOCK_BYTES = 1 << 19
CHAIN_BLOCK_BYTES = 1 << 20
SUBINDEX_WORDS = 64

# every symbol include/starflate_hip.h declares
EXPORTS = [
    "sfh_default_options", "sfh_device_count", "sfh_get_device_props", "sfh_create", "sfh_destroy", "sfh_last_error",
    "sfh_compress_bound", "sfh_compress", "sfh_compress_multi", "sfh_compress_device", "sfh_compress_device_async",
    "sfh_last_block_bytes", "sfh_index_entries", "sfh_copy_index", "sfh_copy_subindex", "sfh_decompress_device", "sfh_decompress", "sfh_last_inflate_ms", "sfh_last_decode_scratch_bytes",
    "sfh_inflate_stage_name", "sfh_checksum_device", "sfh_crc32_combine", "sfh_adler32_combine",
    "sfh_set_profiling", "sfh_last_stage_ms", "sfh_stage_name", "sfh_debug_read",
    "sfh_gather_offsets", "sfh_gather_streams", "sfh_comm_ranks", "sfh_lds_order_check",
]


class Options(C.Structure):
    _fields_ = [("strategy", C.c_uint32), ("final_stream", C.c_uint32), ("lazy", C.c_uint32),
                ("no_stored_fast_path", C.c_uint32), ("container", C.c_uint32), ("block_bytes", C.c_uint32),
                ("effort", C.c_uint32), ("chain_depth", C.c_uint32)]


class DeviceProps(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("arch", C.c_char * 32), ("compute_units", C.c_uint32),
                ("lds_bytes_per_cu", C.c_uint32), ("l2_bytes", C.c_uint32), ("memory_clock_khz", C.c_uint32),
                ("memory_bus_bits", C.c_uint32), ("clock_khz", C.c_uint32), ("total_memory", C.c_uint64)]


_LIB = None


def lib():
    """Load libstarflate_hip.so (loudly failing if it has not been built)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one
    # process cannot both see the GPU.  Import torch first so that the DT_NEEDED
    # libamdhip64.so.7 of our library binds to the copy torch has already loaded (a pure
    # C/C++ host without torch gets /opt/rocm's through the library's RUNPATH).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    L.sfh_default_options.argtypes = [C.POINTER(Options)]
    L.sfh_default_options.restype = None
    L.sfh_device_count.argtypes = []
    L.sfh_device_count.restype = C.c_int
    L.sfh_get_device_props.argtypes = [C.c_int, C.POINTER(DeviceProps)]
    L.sfh_get_device_props.restype = C.c_int
    L.sfh_create.argtypes = [C.POINTER(vp), C.c_int]
    L.sfh_create.restype = C.c_int
    L.sfh_destroy.argtypes = [vp]
    L.sfh_destroy.restype = None
    L.sfh_last_error.argtypes = [vp]
    L.sfh_last_error.restype = C.c_char_p
    L.sfh_compress_bound.argtypes = [sz, C.c_uint32]
    L.sfh_compress_bound.restype = sz
    L.sfh_compress.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(Options)]
    L.sfh_compress.restype = C.c_int
    L.sfh_compress_multi.argtypes = [C.POINTER(vp), C.c_int, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(Options)]
    L.sfh_compress_multi.restype = C.c_int
    L.sfh_compress_device.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(Options), vp]
    L.sfh_compress_device.restype = C.c_int
    L.sfh_compress_device_async.argtypes = [vp, vp, sz, vp, sz, vp, C.POINTER(Options), vp]
    L.sfh_compress_device_async.restype = C.c_int
    L.sfh_gather_offsets.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sfh_gather_offsets.restype = C.c_int
    L.sfh_comm_ranks.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sfh_comm_ranks.restype = C.c_int
    L.sfh_gather_streams.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp]
    L.sfh_gather_streams.restype = C.c_int
    L.sfh_last_block_bytes.argtypes = [vp]
    L.sfh_last_block_bytes.restype = C.c_uint32
    L.sfh_last_decode_scratch_bytes.argtypes = [vp]
    L.sfh_last_decode_scratch_bytes.restype = sz
    L.sfh_index_entries.argtypes = [vp]
    L.sfh_index_entries.restype = sz
    L.sfh_copy_index.argtypes = [vp, vp, sz, C.c_int, vp]
    L.sfh_copy_index.restype = C.c_int
    L.sfh_copy_subindex.argtypes = [vp, vp, sz, C.c_int, vp]
    L.sfh_copy_subindex.restype = C.c_int
    L.sfh_decompress_device.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, C.c_uint32, C.POINTER(C.c_uint32), vp]
    L.sfh_decompress_device.restype = C.c_int
    L.sfh_decompress.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, C.c_uint32, C.POINTER(C.c_uint32)]
    L.sfh_decompress.restype = C.c_int
    L.sfh_last_inflate_ms.argtypes = [vp, C.POINTER(C.c_float * INFLATE_NSTAGES)]
    L.sfh_last_inflate_ms.restype = C.c_int
    L.sfh_inflate_stage_name.argtypes = [C.c_int]
    L.sfh_inflate_stage_name.restype = C.c_char_p
    L.sfh_checksum_device.argtypes = [vp, vp, sz, C.c_uint32, C.POINTER(C.c_uint32), vp]
    L.sfh_checksum_device.restype = C.c_int
    L.sfh_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
    L.sfh_crc32_combine.restype = C.c_uint32
    L.sfh_adler32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
    L.sfh_adler32_combine.restype = C.c_uint32
    L.sfh_set_profiling.argtypes = [vp, C.c_int]
    L.sfh_set_profiling.restype = None
    L.sfh_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float * NSTAGES)]
    L.sfh_last_stage_ms.restype = C.c_int
    L.sfh_stage_name.argtypes = [C.c_int]
    L.sfh_stage_name.restype = C.c_char_p
    L.sfh_debug_read.argtypes = [vp, C.c_int, vp, sz]
    L.sfh_debug_read.restype = C.c_int
    L.sfh_lds_order_check.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sfh_lds_order_check.restype = C.c_int
    _LIB = L
    return L


def device_props(device=0):
    p = DeviceProps()
    rc = lib().sfh_get_device_props(int(device), C.byref(p))
    if rc:
        raise RuntimeError(f"sfh_get_device_props({device}) -> {rc}")
    return {k: (getattr(p, k).decode() if isinstance(getattr(p, k), bytes) else getattr(p, k)) for k, _ in p._fields_ if k != "reserved"}


EFFORT = {"default": 0, "fast": 1, "fastest": 2, "thorough": 3, "max": 4, "best": 5, "ultra": 6, "extreme": 7, "recent": 8, "recent_all": 9}


def make_options(strategy="auto", final_stream=True, lazy=True, stored_fast_path=True, container="raw", block_bytes=0,
                 effort="default"):
    o = Options()
    lib().sfh_default_options(C.byref(o))
    o.strategy = STRATEGY[strategy] if isinstance(strategy, str) else int(strategy)
    o.final_stream = int(bool(final_stream))
    o.lazy = 3 if lazy is True else int(lazy)
    o.no_stored_fast_path = int(not stored_fast_path)
    o.container = CONTAINER[container] if isinstance(container, str) else int(container)
    o.block_bytes = int(block_bytes)
    if isinstance(effort, str) and effort.startswith("chain"):  # "chain4": exact hash chains of that depth
        o.effort, o.chain_depth = EFFORT["best"], int(effort[5:].lstrip(":") or 8)
    else:
        o.effort = EFFORT[effort] if isinstance(effort, str) else int(effort)
    return o


def resolve_block_bytes(block_bytes, n, effort="default"):
    """What sfh_options.block_bytes = 0 stands for on an input of n bytes (mirrors sf_capi.hip): up to 256 KiB, 1 MiB with
    the chain efforts."""
    if block_bytes:
        return int(block_bytes)
    if isinstance(effort, str):  # a name, "chainN", or (as make_options accepts) the enum's integer
        e = EFFORT["best"] if effort.startswith("chain") else EFFORT[effort]
    else:
        e = int(effort)
    chain = EFFORT["best"] <= e <= EFFORT["extreme"]
    b = CHAIN_BLOCK_BYTES if chain else LARGE_BLOCK_BYTES
    while b > DEFAULT_BLOCK_BYTES and n // b < (1024 if chain else 2048):
        b >>= 1
    while b > SEGMENT_BYTES and n // b < 256:
        b >>= 1
    return b
