#!/usr/bin/env python3
"""Development aid: per-kernel ms of the stored fast path (BASELINE config[4]): 256 MiB of high-entropy bytes, and the same with
every fourth 1 MiB compressible (the fast path switching on and off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from starflate_amd import Compressor, synth
n = 256 << 20
g = torch.Generator(device="cuda"); g.manual_seed(5)
rnd = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
mix = rnd.clone()
txt = synth.gen_text_torch(n, seed=3, device="cuda")
for i in range(0, n, 4 << 20):
    mix[i:i + (1 << 20)] = txt[i:i + (1 << 20)]
c = Compressor(0); c.set_profiling(True)
for name, d in (("random", rnd), ("random+text", mix)):
    acc = {}
    for i in range(8):
        out, nb = c.compress_tensor(d)
        if i >= 3:
            for k, v in c.stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 5
    tot = sum(acc.values())
    print(name, {k: round(v, 4) for k, v in acc.items()}, "sum", round(tot, 4), "MiB/s", round(n / tot / 1048.576), "ratio", round(n / nb, 4))
