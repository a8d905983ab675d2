cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_tests.txt
cat gpurun_out/r05_tests.txt
