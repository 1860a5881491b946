# tests, a few timings and the full bench line at the final sources (development aid; GPU box)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/${1:-r5i}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest.log
for e in default recent recent_all; do for w in text source binary; do SF_EFFORT=$e SF_WORKLOAD=$w timeout -k 10 120 python tools/k1_time.py 2>&1 | tail -1 >> $out/time.log; done; done; cat $out/time.log
timeout -k 10 200 python tools/d1_time.py > $out/d1.log 2>&1; cat $out/d1.log
timeout -k 10 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; python - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "ratio_vs_zlib6", "kernel_ms", "roundtrip_ok")})
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "read_frac", "traffic", "kernel_ms")}, d["roofline"]["traffic_info"] and d["roofline"]["traffic_info"]["current"])
print("decompress", {k: (v["ms"], v["value"]) for k, v in d["decompress"].items() if isinstance(v, dict)})
for k, v in d["workloads"].items():
    if isinstance(v, dict) and "value" in v:
        print(f"{k:38s} {v['value']:10.1f} MiB/s  ratio_vs_zlib6 {v['ratio_vs_zlib6']:.4f}  k_lz77 {v['kernel_ms'].get('k_lz77')}")
PY
