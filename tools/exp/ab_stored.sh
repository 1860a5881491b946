# usage: bash tools/exp/ab_stored.sh <out> : build/variants/lib_a.so against lib_b.so, alternating on one box -- the stored fast
# path's kernels on 256 MiB of noise (rand_time.py) and k_lz77 on the text / real bytes (k1_time.py, default effort)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/$1; mkdir -p $out; rm -f $out/ab.log
for lib in a b a b; do
  echo "== $lib" >> $out/ab.log
  SFH_LIB=$PWD/build/variants/lib_$lib.so timeout -k 10 200 python tools/exp/rand_time.py >> $out/ab.log 2>&1 || exit 1
  for w in text source; do echo -n "$lib $w " >> $out/ab.log; SFH_LIB=$PWD/build/variants/lib_$lib.so SF_WORKLOAD=$w timeout -k 10 120 python tools/k1_time.py 2>&1 | tail -1 >> $out/ab.log; done
done
cat $out/ab.log
