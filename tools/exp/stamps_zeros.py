"""Experiment: k_lz77 phase shares (SFH_K1_STAMPS=1) on all-zero input and on a repeated 61-byte line."""
import os, sys
os.environ["SFH_K1_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from starflate_amd import Compressor, _capi
n = 256 << 20
c = Compressor(0)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
line = torch.randint(32, 127, (61,), dtype=torch.uint8, device="cuda", generator=gen)
for name, data in (("zeros", torch.zeros(n, dtype=torch.uint8, device="cuda")), ("period61", line.repeat(n // 61 + 1)[:n].contiguous())):
    for _ in range(2): c.compress_tensor(data)
    both = c.debug(_capi.DBG_STAMPS, n // 32768).astype(np.float64)
    per = c.last_block_bytes() // 32768
    st = both[0][: (n // 32768 + per - 1) // per] / per
    names = ["stage", "match", "take", "walk", "segpre", "emit", "flush"]
    med = np.median(st[:, :7], axis=0)
    print(name, {k: int(v) for k, v in zip(names, med)}, "sum", int(med.sum()), "reconcile rounds", round(float(np.mean(st[:, 7])) / 4, 2))
