#!/bin/bash
# usage: bash tools/prof_extra.sh r06   -> gpurun_out/prof_<tag>x/<tag>_pmc_summary_{recent_all,best,real_source,real_binary}.json (GPU box)
# Counter passes for the kernels and workloads tools/prof_round.sh does not cover (round-5 verdict, item 4): the exact-recency
# and the hash-chain instantiations of k_lz77 on 256 MiB of the bench text, and the default effort on the real-bytes workloads.
# Per variant: a kernel trace, FETCH_SIZE, WRITE_SIZE (separate passes: TCC slots) and one pass of eight SQ counters.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=${1:-r06}
out=gpurun_out/prof_${tag}x
rm -rf $out; mkdir -p $out
python - <<'PY'
import numpy as np
from starflate_amd import realbytes
s = realbytes.source(96 << 20); s[: s.size // (1 << 20) * (1 << 20)].tofile("/tmp/sf_real_source.bin")
b = realbytes.binary(256 << 20); b[: b.size // (1 << 20) * (1 << 20)].tofile("/tmp/sf_real_binary.bin")
PY
BASE="python bench.py --bytes 268435456 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-decompress"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for v in recent_all best real_source real_binary; do
  case $v in
    recent_all) CMD="$BASE --effort recent_all";;
    best) CMD="$BASE --effort best";;
    real_source) CMD="$BASE --input-file /tmp/sf_real_source.bin";;
    real_binary) CMD="$BASE --input-file /tmp/sf_real_binary.bin";;
  esac
  d=$out/$v; mkdir -p $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d/kt -- $CMD > $d/kt.log 2>&1
  bash tools/pmc_run.sh $d/fetch 300 "FETCH_SIZE" -- $CMD > $d/fetch.log 2>&1
  bash tools/pmc_run.sh $d/write 300 "WRITE_SIZE" -- $CMD > $d/write.log 2>&1
  bash tools/pmc_run.sh $d/sq 300 "$SQ" -- $CMD > $d/sq.log 2>&1
  python tools/pmc_summary.py $d/kt $d/fetch $d/write $d/sq > $d/raw.json
  python - $d/raw.json $out/${tag}_pmc_summary_$v.json "$CMD" <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
from starflate_amd.build import source_stamp
d = json.load(open(sys.argv[1]))
out = {"_meta": dict(source_stamp(), command=sys.argv[3], bytes_per_launch=256 << 20,
                     note="sums over the command's dispatches of each kernel (pmc_dispatches); reads = 2 x FETCH_SIZE KiB (gfx950 wide-read correction), "
                          "writes = WRITE_SIZE KiB; SQ_* are wave-instruction / cycle counts as rocprofv3 reports them")}
for k, v in d.items():
    if "k_lz77" in k or k.startswith("k_plan") or k in ("k_emit", "k_scan"):
        n = max(v.get("pmc_dispatches", 1), 1)
        e = dict(v)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["hbm_read_bytes_per_launch"] = int(2 * v["FETCH_SIZE"] * 1024 / n)
            e["hbm_write_bytes_per_launch"] = int(v["WRITE_SIZE"] * 1024 / n)
        if v.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"], 3)
            e["lds_active_cycles_per_lds_instruction"] = round(v["SQ_LDS_IDX_ACTIVE"] / max(v.get("SQ_INSTS_LDS", 1), 1), 2)
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            if c in v:
                e[c + "_per_input_byte"] = round(v[c] / n / (256 << 20), 4)
        out[k] = e
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(sys.argv[2], {k: (round(v.get("trace_avg_us", 0)), v.get("lds_bank_conflict_frac"), v.get("lds_active_cycles_per_lds_instruction")) for k, v in out.items() if k != "_meta"})
PY
done
