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
