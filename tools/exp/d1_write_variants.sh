cd "${GRAFT_REPO_ROOT:-.}"
for v in h0 h3; do
  export SFH_LIB=$PWD/build/variants/lib_$v.so
  rm -rf gpurun_out/r5c/$v; mkdir -p gpurun_out/r5c/$v
  bash tools/pmc_run.sh gpurun_out/r5c/$v 200 "WRITE_SIZE" -- python tools/d1_time.py 268435456 > gpurun_out/r5c/$v.log 2>&1
  python - $v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(f"gpurun_out/r5c/{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_inflate_tokens_sub" in r["Kernel_Name"]:
            acc["w"] += float(r["Counter_Value"]); cnt["w"] += 1
print(sys.argv[1], "k_inflate_tokens_sub WRITE_SIZE per launch (256 MiB):", round(acc["w"] / max(cnt["w"], 1) / 1024, 1), "MiB over", cnt["w"], "launches")
PY
done
