# decoder tests + timing, then the round's profile passes (development aid; GPU box)
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/${1:-r5j}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_inflate.py tests/test_baseline_configs.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout -k 10 200 python tools/d1_time.py > $out/d1.log 2>&1; cat $out/d1.log
bash tools/prof_round.sh r05 > $out/prof.log 2>&1; echo "prof rc $?"; tail -25 $out/prof.log
