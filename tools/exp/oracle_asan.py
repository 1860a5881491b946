"""The oracle (ASan + UBSan build: make -C oracle asan) over the specification modes of round 5 -- recent, chains, the fast path, the
stored-without-a-code rule -- on inputs that reach each of them; run as
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/exp/oracle_asan.py   (CPU; no torch import)"""
import ctypes as C, sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib as O
O._LIB = None
real = C.CDLL
def patched(path, *a, **k):
    if path.endswith("libsf_oracle.so"):
        path = path.replace("libsf_oracle.so", "libsf_oracle_asan.so")
    return real(path, *a, **k)
C.CDLL = patched
import numpy as np, zlib
import importlib.util
def _load(name):
    sp = importlib.util.spec_from_file_location(name, f'starflate_amd/{name}.py'); m = importlib.util.module_from_spec(sp); sp.loader.exec_module(m); return m
synth = _load('synth')
rng = np.random.default_rng(4)
cases = [synth.gen_text(300000, seed=1), rng.integers(0, 256, 200001, dtype=np.uint8), np.zeros(70001, np.uint8), rng.integers(0, 64, 150000, dtype=np.uint8),
         np.concatenate([rng.integers(0, 256, 98304, dtype=np.uint8), synth.gen_text(70000, seed=2), rng.integers(0, 256, 40000, dtype=np.uint8)]),
         np.frombuffer(open('oracle/sf_oracle.c','rb').read()*8, np.uint8)[:400000], np.zeros(0, np.uint8), np.frombuffer(b"abc", np.uint8)]
kws = [dict(), dict(recent=1, near_depth=1, link_steps=1), dict(recent=1, near_depth=3, link_steps=2, stride2=0, step=512), dict(recent=1, near_depth=1, link_steps=1, stride2=0, step=512),
       dict(chain_depth=8), dict(stride2=0, step=512, hash_bits=12, long_hash_bytes=7), dict(fast_skip=0), dict(strategy=3), dict(strip_bytes=32768, recent=1, near_depth=2, link_steps=3)]
for d in cases:
    for kw in kws:
        s = O.compress(d, O.default_params(**kw))
        assert zlib.decompress(bytes(s), -15) == d.tobytes()
        st, w, back = O.decompress(s, d.size)
        assert st == 0 and w == d.size
print("asan/ubsan run ok:", len(cases) * len(kws), "streams")
