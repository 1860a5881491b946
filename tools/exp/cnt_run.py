import os, sys
sys.path.insert(0, os.getcwd())
import torch
from starflate_amd import Compressor, synth
n = 4 << 20
data = synth.gen_text_torch(n, seed=3, device="cuda")
c = Compressor(0)
out, nb = c.compress_tensor(data)
idx, sub, bb = c.last_index(device="cuda"), c.last_subindex(device="cuda"), c.last_block_bytes()
back, st = c.decompress_tensor(out[:nb].clone(), idx, n, subindex=sub, block_bytes=bb)
torch.cuda.synchronize()
print(st, bool(torch.equal(back, data)))
