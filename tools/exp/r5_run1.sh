cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r5b
timeout -k 10 400 python tools/exp/recent_probe.py > gpurun_out/r5b/probe.log 2>&1; echo "probe rc $?" >> gpurun_out/r5b/probe.log
grep -v " OK rt True" gpurun_out/r5b/probe.log | tail -30
for e in default recent recent_all thorough default recent recent_all thorough; do for w in text source binary; do SF_EFFORT=$e SF_WORKLOAD=$w timeout -k 10 120 python tools/k1_time.py 2>&1 | tail -1 >> gpurun_out/r5b/time.log; done; done
cat gpurun_out/r5b/time.log
