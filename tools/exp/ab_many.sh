# usage: bash tools/exp/ab_many.sh <out> <rounds> <lib name...> : k_lz77 (1 GiB of text, default effort) and the stored path's kernels
# (256 MiB of noise) of build/variants/lib_<name>.so, the variants taking turns on one box
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/$1; rounds=$2; shift 2; mkdir -p $out; rm -f $out/ab.log
for k in $(seq $rounds); do for lib in "$@"; do
  echo -n "$lib text " >> $out/ab.log; SFH_LIB=$PWD/build/variants/lib_$lib.so SF_WORKLOAD=text timeout -k 10 120 python tools/k1_time.py 1073741824 2>&1 | tail -1 >> $out/ab.log
  echo -n "$lib " >> $out/ab.log; SFH_LIB=$PWD/build/variants/lib_$lib.so timeout -k 10 200 python tools/exp/rand_time.py 2>&1 | grep "^random " >> $out/ab.log
done; done
cat $out/ab.log
