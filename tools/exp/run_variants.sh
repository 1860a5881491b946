#!/bin/bash
# On the GPU box: time k_lz77 / k_plan / k_emit (tools/k1_time.py) for every build/variants/lib_*.so given by name.
# usage: bash tools/exp/run_variants.sh name1 name2 ...  (appends to gpurun_out/variants.log)
cd "${GRAFT_REPO_ROOT:-.}"
for name in "$@"; do
  echo "== $name" | tee -a gpurun_out/variants.log
  SFH_LIB="$PWD/build/variants/lib_$name.so" timeout -k 10 240 python tools/k1_time.py 2>&1 | tail -1 | tee -a gpurun_out/variants.log
done
