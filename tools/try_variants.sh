#!/bin/bash
# usage: bash tools/try_variants.sh <suffix> ...   (libstarflate_hip_<suffix>.so built beforehand)
cd "${GRAFT_REPO_ROOT:-.}"
make -s -C oracle
for v in "$@"; do
  cp starflate_amd/libstarflate_hip_$v.so starflate_amd/libstarflate_hip.so
  echo "== $v"
  timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'], d['roundtrip_ok'], d['ratio'], d['ratio_vs_zlib6'])"
  python tools/k1_stamps.py 2>/dev/null | tail -2 | head -1
done
