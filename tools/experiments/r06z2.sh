cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -k "flattened or folded or seam or c2_full" > gpurun_out/r06z2_tests.log 2>&1
grep -E "passed|failed|FAILED|^E   +(Assert|assert)" gpurun_out/r06z2_tests.log | head -12
