# usage: bash tools/exp/ab_spec.sh <out> : index-only decoder, speculative wave kernel against the lane-serial one (1 GiB text), then the decoder tests
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/$1; mkdir -p $out; rm -f $out/ab.log
timeout -k 10 400 python -m pytest tests/test_gpu_inflate.py -m gpu -x -v > $out/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit 1
for serial in 0 1; do echo "== SFH_INFLATE_SERIAL=$serial" >> $out/ab.log; SFH_INFLATE_SERIAL=$serial timeout -k 10 200 python tools/d1_time.py 2>&1 | grep -v amdgpu.ids >> $out/ab.log; done
cat $out/ab.log
