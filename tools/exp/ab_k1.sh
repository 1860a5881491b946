# usage: bash tools/exp/ab_k1.sh <out> <effort...> : k_lz77 ms per GiB of build/variants/lib_a.so and lib_b.so, alternating
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/$1; shift; mkdir -p $out; rm -f $out/ab.log
for lib in a b a b; do for e in "$@"; do for w in text source binary; do echo -n "$lib " >> $out/ab.log; SFH_LIB=$PWD/build/variants/lib_$lib.so SF_EFFORT=$e SF_WORKLOAD=$w timeout -k 10 120 python tools/k1_time.py 2>&1 | tail -1 >> $out/ab.log; done; done; done
cat $out/ab.log
