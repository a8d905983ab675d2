cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06ae_tests.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06ae_tests.log | tail -6
