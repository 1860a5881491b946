#!/bin/bash
# usage: bash tools/pmc_run.sh <out_dir> <timeout_s> "<counters>" -- <program> [args...]
# One guarded way to start a rocprofv3 --pmc pass.  It REFUSES a counter set that does not fit one pass on gfx950
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": TCC has 4 slots, FETCH_SIZE costs 3 and WRITE_SIZE 2, SQ has 8):
# round 3 found that two TCC-derived counters in one pass make rocprofv3 abort and then hang until it is killed
# (profiles/r03_notes.md).  The program follows `--` directly (no env / bash -c hop) and runs under `timeout -k`.
set -u
out=$1; lim=$2; counters=$3; shift 3
[ "$1" = "--" ] && shift
tcc=0; sq=0
for c in $counters; do
  case "$c" in
    FETCH_SIZE) tcc=$((tcc + 3));;
    WRITE_SIZE) tcc=$((tcc + 2));;
    TCC_*|TCP_TCC_*) tcc=$((tcc + 1));;
    SQ_*) sq=$((sq + 1));;
  esac
done
if [ $tcc -gt 4 ] || [ $sq -gt 8 ]; then
  echo "pmc_run.sh: refusing '$counters': TCC slots $tcc/4, SQ slots $sq/8 -- split into separate passes" >&2
  exit 64
fi
case "$1" in env|taskset|numactl|bash|sh) echo "pmc_run.sh: the program must follow -- directly, not '$1'" >&2; exit 64;; esac
export TMPDIR=/tmp
mkdir -p "$out"
exec timeout -k 10 "$lim" rocprofv3 --pmc $counters --output-format csv -d "$out" -- "$@"
