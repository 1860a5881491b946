# soak runs of the two fuzz tests at the final sources (development aid; GPU box)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/${1:-r6s}; mkdir -p $out
SF_FUZZ_N=${2:-20000} timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz_bit_exact" > $out/fuzz_c.log 2>&1; echo "compressor fuzz rc $?"; tail -3 $out/fuzz_c.log
SF_FUZZ_N=${5:-1500} timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz_stored" > $out/fuzz_f.log 2>&1; echo "stored fast path fuzz rc $?"; tail -3 $out/fuzz_f.log
SF_FUZZ_N=${3:-5000} timeout -k 10 600 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q -k "fuzz" > $out/fuzz_d.log 2>&1; echo "decoder fuzz rc $?"; tail -3 $out/fuzz_d.log
SF_SPEC_FUZZ=${4:-400} timeout -k 10 900 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q -k "speculative" > $out/fuzz_s.log 2>&1; echo "speculative against lane-serial, damaged streams rc $?"; tail -3 $out/fuzz_s.log
