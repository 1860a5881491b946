#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_lz77 (SFH_K1_STAMPS=1 build; shares only, not a timing claim)."""
import os
import sys

os.environ["SFH_K1_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from starflate_amd import Compressor, _capi, synth

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
data = _input(n)
c = Compressor(0)
for _ in range(2):
    c.compress_tensor(data, block_bytes=bb, effort=os.environ.get("SF_EFFORT", "default"))
both = c.debug(_capi.DBG_STAMPS, n // 32768).astype(np.float64)
per = c.last_block_bytes() // 32768  # k_lz77 stamps one row per strip: scale to a 32 KiB chunk
st = both[0][: (n // 32768 + per - 1) // per] / per
names = ["stage", "match", "take", "walk", "segpre", "emit", "flush"]
med = np.median(st[:, :7], axis=0)
print(os.environ.get("SF_WORKLOAD", "text"), "median cycles per chunk:", {k: int(v) for k, v in zip(names, med)}, "sum", int(med.sum()))
print("reconcile rounds per wave-round (wave 0):", round(float(np.mean(st[:, 7])) * per / (per * 4), 2), "max", float(np.max(both[0][: st.shape[0], 7])) / (per * 4))
print("shares:", {k: round(v / med.sum(), 3) for k, v in zip(names, med)})
k2 = both[1]
names2 = ["load", "ll_lengths", "d_lengths", "costs+rle", "cl_code", "header", "codes+store"]
med2 = np.median(k2[:, :7], axis=0)
print("k_plan median cycles per chunk:", {k: int(v) for k, v in zip(names2, med2)}, "sum", int(med2.sum()))
