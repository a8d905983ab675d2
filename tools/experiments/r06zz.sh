cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06zz_tests.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06zz_tests.log | tail -5
bash tools/experiments/r06zz_profiles.sh > gpurun_out/r06zz_profiles.log 2>&1
tail -12 gpurun_out/r06zz_profiles.log | cut -c1-300
