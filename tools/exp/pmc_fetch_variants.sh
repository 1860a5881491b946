#!/bin/bash
# usage: bash tools/exp/pmc_fetch_variants.sh name...  -> FETCH_SIZE (one TCC counter per pass: two do not fit) of k_lz77 per variant library (256 MiB text)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for name in "$@"; do
  out=gpurun_out/pmcf_$name; rm -rf $out; mkdir -p $out
  export SFH_LIB="$PWD/build/variants/lib_$name.so"
  bash tools/pmc_run.sh $out/p 150 "FETCH_SIZE" -- python bench.py --bytes 268435456 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-decompress > $out/log 2>&1
  python - "$out" "$name" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); cnt = 0
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lz77" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE": cnt += 1
print(sys.argv[2], cnt, {c: round(v / max(cnt, 1) / 1024, 1) for c, v in acc.items()}, "MiB per launch (FETCH x1)")
PY
done
