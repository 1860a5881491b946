#!/bin/bash
# Sensitivity of k_lz77's time to vector instructions added to one phase (round 6).
# HERE:        bash tools/exp/probe_phases.sh build      -> build/variants/lib_{base,stage,match,parse,emit}.so
# on the box:  bash tools/exp/probe_phases.sh run <out>  -> gpurun_out/<out>/probe.log (k_lz77 ms per GiB, variants alternated)
# Every probe adds 128 `v_xor_b32 v, v, v` per wave-round to its phase (match: 14 per interval x 9 intervals = 126).
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  bash tools/exp/build_variants.sh base="" stage="-DSF_PROBE_STAGE=128" match="-DSF_PROBE_MATCH=14" parse="-DSF_PROBE_PARSE=128" emit="-DSF_PROBE_EMIT=128"
  exit
fi
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/$2; mkdir -p $out; rm -f $out/probe.log
for rep in 1 2 3; do for v in base stage match parse emit; do
  echo -n "$v " >> $out/probe.log
  SFH_LIB=$PWD/build/variants/lib_$v.so timeout -k 10 120 python tools/k1_time.py 1073741824 2>&1 | tail -1 >> $out/probe.log
done; done
cat $out/probe.log
