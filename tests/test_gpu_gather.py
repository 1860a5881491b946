"""sfh_gather_streams: the RCCL concatenation of the C-ABI (one process per GPU; SURVEY.md 5: ncclAllGather of the u64 sizes,
then grouped ncclSend / ncclRecv).  A one-GPU box gives a 1-rank communicator: the real RCCL, the real compressor, the
size exchange, the read-back, the root's placement logic; the offsets of N ranks are host arithmetic, tested on CPU
(tests/test_capi_symbols.py), and the N-rank exchange pattern itself is rehearsed over gloo (tests/test_multigpu_gloo.py)."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from starflate_amd import _capi, synth

pytestmark = pytest.mark.gpu


def _rccl():
    import torch

    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    L = C.CDLL(path if os.path.exists(path) else "librccl.so")
    L.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    L.ncclCommInitAll.restype = C.c_int
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    L.ncclCommDestroy.restype = C.c_int
    return L


def test_gather_streams_one_rank_communicator(compressor):
    import torch

    R = _rccl()
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert R.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        lib = compressor._lib
        data = synth.gen_text(21 * 32768 + 77, seed=41)
        src = torch.from_numpy(data).cuda()
        shard = torch.empty(compressor.compress_bound(src.numel()), dtype=torch.uint8, device="cuda")
        size = torch.zeros(1, dtype=torch.int64, device="cuda")
        side = torch.cuda.Stream()
        # the compressor runs on one stream, the gather on another: the ctx orders the gather behind the call
        compressor.compress_tensor_async(src, shard, size, final_stream=True, stream=side.cuda_stream)
        base = 40
        out = torch.zeros(base + shard.numel() + 64, dtype=torch.uint8, device="cuda")
        h_sizes = (C.c_uint64 * 1)()
        end = C.c_uint64(0)
        s = torch.cuda.current_stream().cuda_stream
        rc = lib.sfh_gather_streams(compressor._h, comm, 0, shard.data_ptr(), size.data_ptr(), out.data_ptr(), base, out.numel(),
                                    h_sizes, C.byref(end), C.c_void_p(s))
        assert rc == 0, lib.sfh_last_error(compressor._h)
        torch.cuda.synchronize()
        n = int(h_sizes[0])
        assert n == int(size.item()) and end.value == base + n
        got = out[base:base + n].cpu().numpy()
        assert zlib.decompress(got.tobytes(), -15) == data.tobytes()
        assert not out[:base].any() and not out[base + n:].any()
        assert np.array_equal(got, np.frombuffer(compressor.compress(data), np.uint8))
        # a stream that already lies at its place is not copied (pointer equality): gather in place
        rc = lib.sfh_gather_streams(compressor._h, comm, 0, out.data_ptr() + base, size.data_ptr(), out.data_ptr(), base, out.numel(),
                                    h_sizes, C.byref(end), C.c_void_p(s))
        assert rc == 0 and end.value == base + n
        torch.cuda.synchronize()
        assert np.array_equal(out[base:base + n].cpu().numpy(), got)
        # too small: every rank refuses alike, before any transfer
        rc = lib.sfh_gather_streams(compressor._h, comm, 0, shard.data_ptr(), size.data_ptr(), out.data_ptr(), base, base + n - 1,
                                    h_sizes, C.byref(end), C.c_void_p(s))
        assert rc == -2 and b"do not fit" in lib.sfh_last_error(compressor._h)
        # the root's own stream inside the gathered range of d_out but NOT at its place (a peer's bytes would land on it, or
        # it would be copied onto itself): refused -- by every rank alike, the exchange carries the root's addresses
        for shift in (1, 8, n // 2):
            rc = lib.sfh_gather_streams(compressor._h, comm, 0, out.data_ptr() + base + shift, size.data_ptr(), out.data_ptr(), base,
                                        out.numel(), h_sizes, C.byref(end), C.c_void_p(s))
            assert rc == -1 and b"overlaps" in lib.sfh_last_error(compressor._h), shift
        # ... while one that lies clear of it, before or behind, is fine
        far = torch.zeros(2 * shard.numel() + 64, dtype=torch.uint8, device="cuda")
        far[shard.numel():shard.numel() + n] = out[base:base + n]
        rc = lib.sfh_gather_streams(compressor._h, comm, 0, far.data_ptr() + shard.numel(), size.data_ptr(), far.data_ptr(), 0, n + 5,
                                    h_sizes, C.byref(end), C.c_void_p(s))
        assert rc == 0 and end.value == n
        torch.cuda.synchronize()
        assert np.array_equal(far[:n].cpu().numpy(), got)
        # bad arguments
        assert lib.sfh_gather_streams(compressor._h, comm, 1, shard.data_ptr(), size.data_ptr(), out.data_ptr(), 0, out.numel(),
                                      h_sizes, C.byref(end), C.c_void_p(s)) == -1  # root outside the communicator
        assert lib.sfh_gather_streams(compressor._h, None, 0, shard.data_ptr(), size.data_ptr(), out.data_ptr(), 0, out.numel(),
                                      h_sizes, C.byref(end), C.c_void_p(s)) == -1
    finally:
        R.ncclCommDestroy(comm)
