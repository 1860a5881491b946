"""Ratio of encoder-specification variants (oracle, CPU) against zlib -6 on synthetic AND real bytes.
Usage: python tools/exp/recent_sweep.py [MiB per slice] [variants.py]   (analysis tool; not part of the product)"""
import sys, os, zlib, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle_lib as O
from starflate_amd import synth, realbytes

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = mib << 20
src = realbytes.source()
binb = realbytes.binary()
work = {"text": synth.gen_text(n, seed=3), "mixed": synth.gen_mixed(n, seed=4),
        "srcH": src[:n].copy(), "srcM": src[40 << 20:(40 << 20) + n].copy(), "srcL": src[80 << 20:(80 << 20) + n].copy(),
        "binA": binb[16 << 20:(16 << 20) + n].copy(), "binB": binb[128 << 20:(128 << 20) + n].copy()}
only = os.environ.get("SF_ONLY")
if only:
    work = {k: v for k, v in work.items() if k in only.split(",")}
z6 = {}
for k, d in work.items():
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    z6[k] = len(c.compress(d.tobytes()) + c.flush())
print("zlib-6 ratios: " + "  ".join(f"{k} {d.size / z6[k]:.3f}" for k, d in work.items()), flush=True)
variants = eval(open(sys.argv[2]).read()) if len(sys.argv) > 2 else [("default", dict())]
for name, kw in variants:
    row = []
    t = time.time()
    for k, d in work.items():
        s = O.compress(d, O.default_params(**kw))
        row.append(f"{k} {z6[k] / s.size:.4f}")
    print(f"{name:44s} " + "  ".join(row) + f"  ({time.time() - t:.0f}s)", flush=True)
