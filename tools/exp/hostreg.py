"""Experiment: sfh_compress from pageable caller buffers vs the same buffers pinned with hipHostRegister for the
duration of the call (C-ABI called directly, destination preallocated and touched, so Python copies are not timed)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from starflate_amd import Compressor, _capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
data = synth.gen_text_torch(n, seed=3, device="cuda").cpu().numpy().copy()
c = Compressor(0)
lib, h = c._lib, c._h
cap = c.compress_bound(n)
dst = np.zeros(cap, dtype=np.uint8)
opt = _capi.make_options("auto", True, True, True, "raw", 0, "default")
rt = torch.cuda.cudart()


def run():
    out_n = C.c_size_t(0)
    t = time.perf_counter()
    rc = lib.sfh_compress(h, data.ctypes.data, n, dst.ctypes.data, cap, C.byref(out_n), C.byref(opt))
    dt = time.perf_counter() - t
    assert rc == 0
    return dt * 1e3, out_n.value


def reg(a, nbytes):
    t = time.perf_counter()
    r = rt.cudaHostRegister(a.ctypes.data, nbytes, 0)
    return (time.perf_counter() - t) * 1e3, r


def unreg(a):
    t = time.perf_counter()
    rt.cudaHostUnregister(a.ctypes.data)
    return (time.perf_counter() - t) * 1e3


run()
for trial in range(3):
    p, out = run()
    r_in, _ = reg(data, n)
    a, _ = run()
    r_out, _ = reg(dst, cap)
    b, _ = run()
    u = unreg(data) + unreg(dst)
    # only the bytes the stream can need: a pessimistic slice of dst (n/2) -- what a library-side register could do
    r_in2, _ = reg(data, n)
    r_out2, _ = reg(dst, out + 4096)
    b2, _ = run()
    u2 = unreg(data) + unreg(dst)
    print(f"{n >> 20} MiB -> {out} B | pageable {p:.1f} ms | src pinned {a:.1f} (+reg {r_in:.1f}) | both pinned {b:.1f} "
          f"(+reg {r_in:.1f}+{r_out:.1f}, unreg {u:.1f}) | both, dst exact {b2:.1f} (+reg {r_in2:.1f}+{r_out2:.1f}, unreg {u2:.1f})")
