"""Ratio of encoder-spec variants (oracle, CPU) against zlib -6 on the synthetic workloads.
Usage: python tools/exp/ratio_sweep.py [MiB]   (analysis tool; not part of the product)"""
import sys, os, zlib, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle_lib as O
from starflate_amd import synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = mib << 20
def gen_runs(n):  # bench.py's "runs" workload: half zeros, half one 61-byte line repeated
    rng = np.random.default_rng(9)
    line = rng.integers(32, 127, 61, dtype=np.uint8)
    d = np.zeros(n, np.uint8)
    d[n // 2:] = np.tile(line, (n - n // 2) // 61 + 1)[: n - n // 2]
    return d


work = {"text": synth.gen_text(n, seed=3), "mixed": synth.gen_mixed(n, seed=4), "runs": gen_runs(n)}
z6 = {}
for k, d in work.items():
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    z6[k] = len(c.compress(d.tobytes()) + c.flush())

variants = eval(open(sys.argv[2]).read()) if len(sys.argv) > 2 else [
    ("base 32K indep", dict()),
    ("strip 256K", dict(strip_bytes=262144)),
]
for name, kw in variants:
    row = []
    for k, d in work.items():
        t = time.time()
        s = O.compress(d, O.default_params(**kw))
        row.append(f"{k} {z6[k] / s.size:.4f}")
    print(f"{name:40s} " + "  ".join(row), flush=True)
