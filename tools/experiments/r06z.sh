cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r06z_tests.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06z_tests.log | tail -3
bash tools/experiments/r06z_profiles.sh > gpurun_out/r06z_profiles.log 2>&1
tail -12 gpurun_out/r06z_profiles.log | cut -c1-300
