#!/bin/bash
# usage: bash tools/prof_round.sh r01   -> profiles/<tag>_* (run on the GPU box; copy back via gpurun_out/)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r01}
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $CMD > $out/kt.log 2>&1
bash tools/pmc_run.sh $out/pmc_fetch 400 "FETCH_SIZE" -- $CMD > $out/pmc_fetch.log 2>&1
bash tools/pmc_run.sh $out/pmc_write 400 "WRITE_SIZE" -- $CMD > $out/pmc_write.log 2>&1
bash tools/pmc_run.sh $out/pmc1 400 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" -- $CMD > $out/pmc1.log 2>&1
bash tools/pmc_run.sh $out/pmc2 400 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" -- $CMD > $out/pmc2.log 2>&1
python tools/pmc_summary.py $out/kt $out/pmc_fetch $out/pmc_write $out/pmc1 $out/pmc2 > $out/${tag}_pmc_summary.json
# kernel stats restricted to this library's kernels (the rest is torch's data generator)
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
(head -1 $f; grep 'sf::' $f) > $out/${tag}_kernel_stats.csv
cat $out/${tag}_kernel_stats.csv
grep '^{"metric"' $out/kt.log | tail -1 > $out/${tag}_bench_under_rocprof.json
python tools/pmc_traffic.py $out/${tag}_pmc_summary.json $out/pmc_traffic.json ${tag}_pmc_summary.json > /dev/null
# BASELINE config[4] (high-entropy input, stored fast path): HBM bytes moved per launch against 2N + 5 per chunk.
# (The decompress leg is off: every kernel counted here is the compressor's.)
CMDR="python bench.py --workload random --bytes 268435456 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-decompress"
bash tools/pmc_run.sh $out/rnd_fetch 300 "FETCH_SIZE" -- $CMDR > $out/rnd_fetch.log 2>&1
bash tools/pmc_run.sh $out/rnd_write 300 "WRITE_SIZE" -- $CMDR > $out/rnd_write.log 2>&1
python tools/pmc_summary.py $out/rnd_fetch $out/rnd_write > $out/rnd_summary.json
python - $out/rnd_summary.json $out/${tag}_random_traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
n = 256 << 20
out = {"_meta": {"workload": "256 MiB high-entropy bytes (bench.py --workload random), default effort", "algorithmic_bytes": 2 * n + 5 * (n >> 15),
                 "note": "reads = 2 x FETCH_SIZE KiB (gfx950 wide-read correction), writes = WRITE_SIZE KiB; per launch = sum / dispatches"}}
tot = 0
for k, v in d.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    m = v.get("pmc_dispatches", 1)
    rd, wr = 2 * v["FETCH_SIZE"] * 1024 / m, v["WRITE_SIZE"] * 1024 / m
    out[k.replace("void ", "").split("<")[0]] = {"read_bytes": int(rd), "write_bytes": int(wr), "dispatches": m}
    tot += rd + wr
out["_meta"]["hbm_bytes_per_step"] = int(tot)
out["_meta"]["over_input_bytes"] = round(tot / n, 3)
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps(out["_meta"]))
PY
