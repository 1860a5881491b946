"""CPU-side checks of the drop-in boundary: libstarflate_hip.so builds for gfx950, loads, and
exports every symbol include/starflate_hip.h declares; no compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from starflate_amd import _capi, build


def _declared():
    with open(os.path.join(ROOT, "include", "starflate_hip.h")) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(sfh_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build()
    lib = _capi.lib()
    declared = _declared()
    assert declared, "no declarations parsed"
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_capi.EXPORTS) == declared


def test_host_side_entry_points_without_gpu():
    lib = _capi.lib()
    o = _capi.Options()
    lib.sfh_default_options(C.byref(o))
    assert (o.strategy, o.final_stream, o.lazy, o.no_stored_fast_path, o.container, o.block_bytes, o.effort) == (0, 1, 3, 0, 0, 0, 0)
    assert o.chain_depth == 0 and C.sizeof(o) == 32
    for bb in (0, 32768, 262144):  # sfh_compress_bound(n, block_bytes); per 32 KiB DEFLATE block
        assert lib.sfh_compress_bound(0, bb) == 32768 + 4096 + 640
        assert lib.sfh_compress_bound(32768, bb) == 32768 + 4096 + 640
        assert lib.sfh_compress_bound(32769, bb) == 2 * (32768 + 4096 + 640)
    for bb in (1000, 32768 + 1, (16 << 20) + 32768):  # what sfh_compress rejects has no bound
        assert lib.sfh_compress_bound(32768, bb) == 0
    assert lib.sfh_compress_bound(1 << 20, 16 << 20) == 32 * (32768 + 4096 + 640)
    assert _capi.resolve_block_bytes(0, 1 << 30) == 524288 and _capi.resolve_block_bytes(0, 512 << 20) == 262144 and _capi.resolve_block_bytes(0, 1 << 20) == 32768
    import oracle_lib as O  # the specification's rule for strip_bytes = 0 is the library's for block_bytes = 0

    for n in (0, 1, 32768, 8 << 20, (8 << 20) + 1, 16 << 20, 64 << 20, (64 << 20) - 1, 1 << 30, 5 << 30):
        assert _capi.resolve_block_bytes(0, n) == O.resolve_strip_bytes(O.default_params(), n), n
        for eff, depth in (("best", 8), ("extreme", 32), ("chain4", 4)):  # the chain efforts: up to 1 MiB
            assert _capi.resolve_block_bytes(0, n, eff) == O.resolve_strip_bytes(O.default_params(chain_depth=depth), n), (n, eff)
    assert _capi.resolve_block_bytes(0, 1 << 30, "ultra") == 1 << 20 and _capi.resolve_block_bytes(0, 256 << 20, "best") == 262144
    # sfh_gather_offsets: where every rank's stream lands on the root (host arithmetic of sfh_gather_streams)
    sizes, off = (C.c_uint64 * 4)(10, 0, 7, 3), (C.c_uint64 * 5)()
    assert lib.sfh_gather_offsets(sizes, 4, 5, 25, off) == 0 and list(off) == [5, 15, 15, 22, 25]
    assert lib.sfh_gather_offsets(sizes, 4, 5, 24, off) == -2  # SFH_E_DST_TOO_SMALL, the same on every rank
    assert lib.sfh_gather_offsets(sizes, 1, 0, 10, off) == 0 and list(off)[:2] == [0, 10]
    big = (C.c_uint64 * 2)(2**64 - 1, 5)
    assert lib.sfh_gather_offsets(big, 2, 1, 2**64 - 1, off) == -1  # overflow
    assert lib.sfh_gather_offsets(None, 2, 0, 0, off) == -1 and lib.sfh_gather_offsets(sizes, 0, 0, 0, off) == -1
    assert _capi.EFFORT == {"default": 0, "fast": 1, "fastest": 2, "thorough": 3, "max": 4, "best": 5, "ultra": 6, "extreme": 7,
                            "recent": 8, "recent_all": 9}  # enum sfh_effort
    with open(os.path.join(ROOT, "include", "starflate_hip.h")) as f:  # ... as the header spells it
        enum = re.search(r"enum sfh_effort \{(.*?)\}", f.read(), flags=re.S).group(1)
    assert {k.strip().replace("SFH_EFFORT_", "").lower(): int(v) for k, v in (e.split("=") for e in enum.split(","))} == _capi.EFFORT
    # an effort may be given by name, as "chainN", or as the enum's integer: the strip rule must not depend on the spelling
    for n in (1 << 30, 600 << 20, 64 << 20):
        for name, val in _capi.EFFORT.items():
            assert _capi.resolve_block_bytes(0, n, name) == _capi.resolve_block_bytes(0, n, val), (n, name)
    assert _capi.resolve_block_bytes(0, 1 << 30, 6) == 1 << 20 and _capi.resolve_block_bytes(0, 1 << 30, 8) == 524288
    assert _capi.resolve_block_bytes(0, 1 << 30, "recent_all") == 524288
    assert lib.sfh_stage_name(0) == b"k_lz77" and lib.sfh_stage_name(3) == b"k_emit" and lib.sfh_stage_name(9) == b""
    import zlib
    a, b = bytes(range(256)) * 300, b"starflate" * 5000  # host-side checksum combine rules against zlib
    assert lib.sfh_crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert lib.sfh_adler32_combine(zlib.adler32(a), zlib.adler32(b), len(b)) == zlib.adler32(a + b)
    assert lib.sfh_crc32_combine(zlib.crc32(a), zlib.crc32(b""), 0) == zlib.crc32(a)


def test_no_cpu_fallback():
    """Without a HIP device the product raises; it never routes to the oracle or any CPU path."""
    lib = _capi.lib()
    if lib.sfh_device_count() > 0:
        pytest.skip("a GPU is present")
    from starflate_amd import Compressor, StarflateError, compress

    with pytest.raises(StarflateError):
        Compressor(0)
    with pytest.raises(StarflateError):
        compress(b"abc")
    h = C.c_void_p()
    assert lib.sfh_create(C.byref(h), 0) == -3 and not h.value


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "starflate_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                with open(os.path.join(dirpath, fn), errors="ignore") as f:
                    txt = f.read()
                assert "oracle_lib" not in txt and "sf_oracle" not in txt and "sfo_" not in txt, fn


def test_item_format_constants_match_the_python_decoder():
    """k_lz77 -> k_emit items (sf_device.h: kItemTok, kItemHead, kItemRegion, kItemRegionShift; round 6: every item says what it is)
    are decoded by Compressor.debug_tokens for the stage-parity tests: the two spellings of the format must not drift apart."""
    import inspect

    from starflate_amd import compressor

    with open(os.path.join(ROOT, "starflate_amd", "csrc", "sf_device.h")) as f:
        h = f.read()
    c = {k: int(v, 0) for k, v in re.findall(r"constexpr uint32_t (kItem\w+) = (0x[0-9A-Fa-f]+|\d+)u?;", h)}
    assert c == {"kItemTok": 0x8000, "kItemRegion": 0x4000, "kItemRegionShift": 9, "kItemHead": 0x0100, "kItemsSkipped": 0x80000000}, c
    assert c["kItemHead"] == 1 << 8 and c["kItemRegionShift"] == 9  # a token's low nine bits index k_emit's table; the region index sits above them
    src = inspect.getsource(compressor.Compressor.debug_tokens)
    for needle in ("(it & 0x8000) != 0", "(it & 0x0100) != 0", "(it & 0x4000) != 0", "(it >> 9) & 31"):
        assert needle in src, needle
    with open(os.path.join(ROOT, "include", "starflate_hip.h")) as f:
        doc = f.read()
    assert "0x8000 | byte" in doc and "0x8100 | len-3" in doc and "bits 9..13" in doc
