import sys, os, zlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, oracle_lib as O
from starflate_amd import Compressor
data = np.load(os.path.join(os.path.dirname(__file__), "it80.npy"))
c = Compressor(0)
for strat, sid in (("fixed", 2), ("auto", 0), ("dynamic", 3)):
    for bb in (131072, 32768):
        own = np.frombuffer(c.compress(data, strategy=strat, block_bytes=bb), np.uint8)
        idx, sub = c.last_index(), c.last_subindex()
        want = O.compress(data, O.default_params(strategy=sid, strip_bytes=bb))
        st, w, back = O.decompress(own, data.size)
        print(strat, bb, "n", data.size, "gpu", own.size, "oracle", want.size, "equal", np.array_equal(own, want), "oracle-decode", st, w,
              "idx", idx.tolist())
        for s in (None, sub):
            got, st2 = c.decompress(own, idx, data.size, subindex=s, block_bytes=bb)
            print("   gpu decode sub=", s is not None, "status", st2, got == data.tobytes())
