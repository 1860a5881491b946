import sys, os, zlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib as O
from starflate_amd import Compressor, synth
c = Compressor(0)
data = np.concatenate([synth.gen_text(20 << 20, seed=71), synth.gen_mixed(13 << 20, seed=72, stripe=1 << 18)[: (13 << 20) - 77]])
for bb in (16 << 20, 4 << 20, 3 * 32768):
    got = np.frombuffer(c.compress(data, block_bytes=bb), np.uint8)
    want = O.compress(data, O.default_params(strip_bytes=bb))
    ok = np.array_equal(got, want)
    rt = zlib.decompress(bytes(got), -15) == data.tobytes()
    back, st = c.decompress(got, c.last_index(), data.size, subindex=c.last_subindex(), block_bytes=bb)
    print(bb, got.size, want.size, ok, rt, st, back == data.tobytes())
