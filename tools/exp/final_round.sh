# usage: bash tools/exp/final_round.sh <out> [tag]: parity + profile passes + the full bench line at the final sources, in this order (GPU box)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/${1:-r6z}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -3 $out/pytest.log
bash tools/prof_round.sh ${2:-r06} > $out/prof.log 2>&1; echo "prof rc $?"; tail -12 $out/prof.log
cp gpurun_out/prof_${2:-r06}/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; python - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "ratio_vs_zlib6", "kernel_ms", "roundtrip_ok")})
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "read_frac", "traffic", "kernel_ms")}, d["roofline"]["traffic_info"] and d["roofline"]["traffic_info"]["current"])
print("decompress", {k: (v["ms"], v["value"]) for k, v in d["decompress"].items() if isinstance(v, dict)})
print("line bytes", len(open(sys.argv[1]).read().strip().splitlines()[-1]))
for k, v in list(d["configs"].items()) + list((d.get("real_bytes") or {}).items()):
    print(f"{k:38s} {v['value']:10.1f} MiB/s  ratio_vs_zlib6 {v['ratio_vs_zlib6']:.4f}  {v.get('kernel_ms')}  {v.get('roofline')}")
PY
