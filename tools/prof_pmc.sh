#!/bin/bash
# usage: bash tools/prof_pmc.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/summary.json
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
ARGS="--bytes 268435456 --steps 2 --warmup 1 --no-cpu-baseline --no-verify $@"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python bench.py $ARGS > $out/kt.log 2>&1
bash tools/pmc_run.sh $out/pmc1 300 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" -- python bench.py $ARGS > $out/pmc1.log 2>&1
bash tools/pmc_run.sh $out/pmc2 300 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" -- python bench.py $ARGS > $out/pmc2.log 2>&1
python tools/pmc_summary.py $out/kt $out/pmc1 $out/pmc2 > $out/summary.json
cat $out/summary.json
