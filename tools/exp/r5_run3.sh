cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5d
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5d/pytest.log
tail -6 gpurun_out/r5d/pytest.log
python - > gpurun_out/r5d/random.log 2>&1 <<'PY'
import torch, time
from starflate_amd import Compressor
c = Compressor(0); c.set_profiling(True)
n = 256 << 20
g = torch.Generator(device="cuda"); g.manual_seed(5)
d = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty(c.compress_bound(n), dtype=torch.uint8, device="cuda")
for i in range(3): c.compress_tensor(d, out=out)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(5): _, nb = c.compress_tensor(d, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print("random 256MiB", round(n / dt / 2**20), "MiB/s", round(dt * 1e3, 3), "ms", {k: round(v, 4) for k, v in c.stage_ms().items()}, nb)
PY
cat gpurun_out/r5d/random.log
for e in default recent; do for w in text binary; do SF_EFFORT=$e SF_WORKLOAD=$w timeout -k 10 120 python tools/k1_time.py 2>&1 | tail -1 >> gpurun_out/r5d/time.log; done; done; cat gpurun_out/r5d/time.log
