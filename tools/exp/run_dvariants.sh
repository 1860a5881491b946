#!/bin/bash
# On the GPU box: decoder kernel ms (tools/d1_time.py) for every build/variants/lib_*.so given by name.
# usage: bash tools/exp/run_dvariants.sh name1 name2 ...  (appends to gpurun_out/dvariants.log)
cd "${GRAFT_REPO_ROOT:-.}"
for name in "$@"; do
  echo "== $name" | tee -a gpurun_out/dvariants.log
  SFH_LIB="$PWD/build/variants/lib_$name.so" timeout -k 10 240 python tools/d1_time.py 2>&1 | grep "^sub\|^idx" | tee -a gpurun_out/dvariants.log
done
