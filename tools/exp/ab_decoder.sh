# usage: bash tools/exp/ab_decoder.sh <out> : decoder kernel ms of build/variants/lib_a.so and lib_b.so, alternating (1 GiB text)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/$1; mkdir -p $out; rm -f $out/ab.log
for lib in a b a b; do echo "== $lib" >> $out/ab.log; SFH_LIB=$PWD/build/variants/lib_$lib.so timeout -k 10 200 python tools/d1_time.py 2>&1 | grep -v amdgpu.ids >> $out/ab.log; done
cat $out/ab.log
SFH_LIB=$PWD/build/variants/lib_b.so timeout -k 10 600 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q 2>&1 | tail -2
