"""Experiment: where the one-rank rehearsal of the N > 1 path (bench.py --force-dist) loses time against the plain path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from starflate_amd import Compressor, synth, multigpu, _capi
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 1 << 30
data = synth.gen_text_torch(n, seed=3, device=dev)
comp = Compressor(0)
bb = _capi.resolve_block_bytes(0, n)
def run(K, label, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(f"{label:40s} {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
out = torch.empty(comp.compress_bound(n), dtype=torch.uint8, device=dev)
run(1, "plain compress_tensor", lambda: comp.compress_tensor(data, out=out, block_bytes=bb))
szd = torch.zeros(1, dtype=torch.int64, device=dev)
def asyn():
    comp.compress_tensor_async(data, out, szd, block_bytes=bb); return int(szd.item())
run(1, "async + size read", asyn)
for K in (1, 4):
    pieces = list(data.chunk(K)); bound = comp.compress_bound(n // K)
    scratch = [torch.empty(bound, dtype=torch.uint8, device=dev) for _ in range(K)]
    size_dev = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(K)]
    gathered = torch.empty(bound * K + 64, dtype=torch.uint8, device=dev)
    def cf(piece, final, k):
        comp.compress_tensor_async(piece, scratch[k], size_dev[k], final_stream=final, block_bytes=bb); return scratch[k], size_dev[k]
    run(K, f"compress_pipelined K={K}", lambda: multigpu.compress_pipelined(cf, pieces, out=gathered))
    run(K, f"compress_pipelined K={K} validate=False", lambda: multigpu.compress_pipelined(cf, pieces, out=gathered, validate=False))
    def seq():
        for k in range(K): comp.compress_tensor_async(pieces[k], scratch[k], size_dev[k], final_stream=(k == K - 1), block_bytes=bb)
        return [int(s.item()) for s in size_dev]
    run(K, f"K={K} async calls only", seq)
dist.destroy_process_group()
