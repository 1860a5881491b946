"""Experiment: GPU decoder kernel times on a fixed-Huffman stream (no code longer than the one-read tables) against
the dynamic one -- how much the long-code path costs the region lanes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from starflate_amd import Compressor, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
data = synth.gen_text_torch(n, seed=3, device="cuda")
c = Compressor(0)
c.set_profiling(True)
for strategy in ("auto", "fixed"):
    out, nb = c.compress_tensor(data, strategy=strategy)
    idx, sub, bb = c.last_index(device="cuda"), c.last_subindex(device="cuda"), c.last_block_bytes()
    stream = out[:nb].clone()
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        _, st = c.decompress_tensor(stream, idx, n, out=back, subindex=sub, block_bytes=bb)
    print(strategy, nb, st, {k: round(v, 3) for k, v in c.inflate_ms().items()}, bool(torch.equal(back, data)))
