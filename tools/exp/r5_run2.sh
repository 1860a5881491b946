cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5c
bash tools/exp/r5_ab.sh r5c recent recent_all > /dev/null 2>&1
timeout -k 10 300 python tools/exp/recent_probe.py > gpurun_out/r5c/probe.log 2>&1; echo "probe rc $?" >> gpurun_out/r5c/probe.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q -k "stored or config4 or fuzz or all_literal or stage_parity or bench_generators" > gpurun_out/r5c/pytest.log 2>&1
python - > gpurun_out/r5c/random.log 2>&1 <<'PY'
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
cat gpurun_out/r5c/ab.log; grep -c " OK rt True" gpurun_out/r5c/probe.log; grep -v " OK rt True" gpurun_out/r5c/probe.log | tail -5; tail -5 gpurun_out/r5c/pytest.log; cat gpurun_out/r5c/random.log
