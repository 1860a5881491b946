#!/usr/bin/env python3
"""Development aid: k_lz77 / k_plan / k_emit milliseconds on 256 MiB of the bench text (HIP events), scaled to 1 GiB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starflate_amd import Compressor, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
bb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
effort = os.environ.get("SF_EFFORT", "default")
data = synth.gen_text_torch(n, seed=3, device="cuda")
c = Compressor(0)
c.set_profiling(True)
acc = {}
for i in range(6):
    out, nb = c.compress_tensor(data, block_bytes=bb, effort=effort)
    if i >= 2:
        for k, v in c.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 4
print(effort, {k: round(v * (1 << 30) / n, 3) for k, v in acc.items()}, "ratio", round(n / nb, 4))
