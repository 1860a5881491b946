#!/bin/bash
# usage: bash tools/exp/pmc_variants.sh name...  -> SQ_INSTS_VALU etc. of k_lz77 per variant library (256 MiB text)
cd "${GRAFT_REPO_ROOT:-.}"
for name in "$@"; do
  echo "== $name"
  SFH_LIB="$PWD/build/variants/lib_$name.so" bash tools/exp/pmc_quick.sh v_$name --no-verify 2>&1 | grep lz77
done
