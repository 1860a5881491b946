#!/bin/bash
# PMC counters of the decoder kernels (development aid): bash tools/exp/d1_pmc.sh
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out/d1pmc; rm -rf $out; mkdir -p $out
bash tools/pmc_run.sh $out/p1 300 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" -- python tools/d1_time.py > $out/p1.log 2>&1
bash tools/pmc_run.sh $out/p2 300 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" -- python tools/d1_time.py > $out/p2.log 2>&1
python - <<'P'
import csv,glob,collections,re
for d in ('p1','p2'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(f'gpurun_out/d1pmc/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            m=re.search(r'k_inflate_\w+', r['Kernel_Name'])
            if not m: continue
            k=m.group(0)
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        print(d,k,{c: round(x/cnt[(k,c)]/1e6,1) for c,x in v.items()})
P
