#!/bin/bash
# usage: bash tools/exp/pmc_quick.sh <tag> [bench args]  -> instruction counts of the library's kernels (one PMC pass)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmcq_$tag
rm -rf $out; mkdir -p $out
bash tools/pmc_run.sh $out/p 300 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" -- python bench.py --bytes 268435456 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-decompress "$@" > $out/log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "sf::" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k in acc:
    print(k, cnt[k], {c: round(v / max(cnt[k], 1) / 1e6, 2) for c, v in sorted(acc[k].items())})
PY
