"""Experiment: k_plan cycles per phase and chunk (SFH_K1_STAMPS=1) on the bench text, real source text and machine code."""
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
names = ["load", "ll lengths", "d lengths", "costs+rle", "cl code", "header bits", "codes+stores"]
for name, data in sets:
    n = data.numel()
    for _ in range(2): c.compress_tensor(data)
    both = c.debug(_capi.DBG_STAMPS, n // 32768).astype(np.float64)
    st = both[1][: n // 32768]
    med = np.mean(st[:, :7], axis=0)
    print(f"{name:7s}", {k: int(v) for k, v in zip(names, med)}, "sum", int(med.sum()), flush=True)
