set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout -k 10 400 python bench.py > gpurun_out/bench_1g.log 2>&1
tail -1 gpurun_out/bench_1g.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1
find gpurun_out/prof/kt -name '*stats*' | head
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/prof/pmc1 -- python bench.py --bytes 268435456 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > gpurun_out/prof_pmc1.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/prof/pmc2 -- python bench.py --bytes 268435456 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > gpurun_out/prof_pmc2.log 2>&1
ls -R gpurun_out/prof | head -40
