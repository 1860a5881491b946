#!/bin/bash
# usage: bash tools/exp/run_ab.sh <variant> -> k_lz77 ms per GiB of build/variants/lib_base.so and lib_<variant>.so, alternating, on
# text / source / machine code; then the parity suites with the working tree's library
cd "${GRAFT_REPO_ROOT:-.}"
v=$1; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/$v.log
for lib in base $v base $v; do for w in text source binary; do echo "== $lib $w" >> gpurun_out/ab/$v.log; SFH_LIB=$PWD/build/variants/lib_$lib.so SF_WORKLOAD=$w python tools/k1_time.py 2>&1 | tail -1 >> gpurun_out/ab/$v.log; done; done
cat gpurun_out/ab/$v.log
python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q 2>&1 | tail -2
