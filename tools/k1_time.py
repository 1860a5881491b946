#!/usr/bin/env python3
"""Development aid: k_lz77 / k_plan / k_emit milliseconds on 256 MiB of the bench text (HIP events), scaled to 1 GiB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starflate_amd import Compressor, synth
def _input(n):
    """SF_WORKLOAD = text (default) | mixed | random | source | binary: the bench generators or the real bytes of
    starflate_amd/realbytes.py (tiled to n: the window is 32 KiB and strips are independent, so tiling a corpus far longer
    than that changes neither the parse nor the ratio)."""
    import numpy as np
    from starflate_amd import realbytes
    w = os.environ.get("SF_WORKLOAD", "text")
    if w == "text":
        return synth.gen_text_torch(n, seed=3, device="cuda")
    if w == "mixed":
        return torch.from_numpy(synth.gen_mixed(n, seed=4)).cuda()
    if w == "random":
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        return torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
    buf = realbytes.source(96 << 20) if w == "source" else realbytes.binary(min(n, 256 << 20))
    return torch.from_numpy(np.resize(buf, n)).cuda()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
bb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
effort = os.environ.get("SF_EFFORT", "default")
data = _input(n)
c = Compressor(0)
c.set_profiling(True)
acc = {}
for i in range(6):
    out, nb = c.compress_tensor(data, block_bytes=bb, effort=effort)
    if i >= 2:
        for k, v in c.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 4
print(os.environ.get("SF_WORKLOAD", "text"), effort, {k: round(v * (1 << 30) / n, 3) for k, v in acc.items()}, "ratio", round(n / nb, 4))
