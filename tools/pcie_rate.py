#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point sfh_compress (H2D + kernels + D2H, pageable host memory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starflate_amd import Compressor, synth
n = 256 << 20
data = synth.gen_text_torch(n, seed=3, device="cuda").cpu().numpy()
c = Compressor(0)
c.compress(data[: 1 << 20])
t = time.perf_counter(); out = c.compress(data); dt = time.perf_counter() - t
t = time.perf_counter(); out = c.compress(data); dt = min(dt, time.perf_counter() - t)
print(f"sfh_compress host->host: {n / dt / 2**20:.0f} MiB/s ({n >> 20} MiB in {dt * 1e3:.1f} ms, ratio {n / len(out):.3f})")
