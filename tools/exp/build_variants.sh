#!/bin/bash
# Build side-by-side variants of the library for kernel experiments (run HERE, before gpurun; the .so files travel).
# usage: bash tools/exp/build_variants.sh name1="-DFLAG=1 ..." name2="..."   -> build/variants/lib_<name>.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/variants
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  ( SF_HIPCC_FLAGS="$flags" SFH_LIB="$PWD/build/variants/lib_$name.so" python -c "from starflate_amd import build as b; b.build(force=True)" && echo "built $name [$flags]" ) &
done
wait
