"""Experiment: k_lz77 phase shares (SFH_K1_STAMPS=1, cycles per 32 KiB chunk) on the bench text, real source text and real machine code."""
import os, sys
os.environ["SFH_K1_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from starflate_amd import Compressor, _capi, synth, realbytes
c = Compressor(0)
sets = [("text", synth.gen_text_torch(256 << 20, seed=3, device="cuda"))]
for name, fn in (("source", realbytes.source), ("binary", realbytes.binary)):
    b = fn()
    if b is not None and len(b) >= (32 << 20):
        a = np.frombuffer(b, np.uint8)[: (len(b) // 262144) * 262144]
        sets.append((name, torch.from_numpy(a.copy()).cuda()))
efforts = sys.argv[1:] or ["default", "recent_all"]
for name, data in sets:
    n = data.numel()
    for effort in efforts:
        for _ in range(2): out, nb = c.compress_tensor(data, effort=effort)
        both = c.debug(_capi.DBG_STAMPS, n // 32768).astype(np.float64)
        per = c.last_block_bytes() // 32768
        st = both[0][: (n // 32768 + per - 1) // per] / per
        names = ["stage", "match", "take", "walk", "segpre", "emit", "flush"]
        med = np.mean(st[:, :7], axis=0)
        print(f"{name:7s} {effort:10s} ratio {n / nb:.3f}", {k: int(v) for k, v in zip(names, med)}, "sum", int(med.sum()), flush=True)
