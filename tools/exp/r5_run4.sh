cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5f
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q -k "stored or config4 or fuzz or bit_exact_vs_oracle or bench_generators" > gpurun_out/r5f/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5f/pytest.log
tail -4 gpurun_out/r5f/pytest.log
python - > gpurun_out/r5f/random.log 2>&1 <<'PY'
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
d64 = torch.randint(0, 64, (n,), dtype=torch.uint8, device="cuda", generator=g)
for i in range(3): c.compress_tensor(d64, out=out)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(5): _, nb = c.compress_tensor(d64, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print("base64ish 256MiB", round(n / dt / 2**20), "MiB/s", round(dt * 1e3, 3), "ms", {k: round(v, 4) for k, v in c.stage_ms().items()}, nb)
PY
cat gpurun_out/r5f/random.log
for e in default default; do for w in text source; do SF_EFFORT=$e SF_WORKLOAD=$w timeout -k 10 120 python tools/k1_time.py 2>&1 | tail -1 >> gpurun_out/r5f/time.log; done; done; cat gpurun_out/r5f/time.log

