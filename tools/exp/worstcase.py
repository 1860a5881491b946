"""Experiment: k_lz77 / whole-path time on degenerate inputs (all zeros, short periods, one repeated line)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from starflate_amd import Compressor, synth
n = 256 << 20
c = Compressor(0); c.set_profiling(True)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
line = torch.randint(32, 127, (61,), dtype=torch.uint8, device="cuda", generator=gen)
inputs = {
    "text": synth.gen_text_torch(n, seed=3, device="cuda"),
    "zeros": torch.zeros(n, dtype=torch.uint8, device="cuda"),
    "period7": (torch.arange(n, device="cuda") % 7).to(torch.uint8),
    "period61 line": line.repeat(n // 61 + 1)[:n].contiguous(),
    "period4096": torch.randint(0, 256, (4096,), dtype=torch.uint8, device="cuda", generator=gen).repeat(n // 4096),
}
for name, data in inputs.items():
    for _ in range(3):
        out, nb = c.compress_tensor(data)
    ms = c.stage_ms()
    tot = sum(ms.values())
    print(f"{name:14s} ratio {n/nb:9.2f}  " + " ".join(f"{k}={v*(1<<30)/n:.2f}" for k, v in ms.items()) + f"  -> {n/2**20/(tot/1e3)/1e3:.1f} K MiB/s")
