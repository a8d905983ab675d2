cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or c2_full or plans" > gpurun_out/r04ah_test.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04ah_test.txt
bash tools/experiments/ab_bench.sh base ring > gpurun_out/r04ah_ab.txt 2>&1; cat gpurun_out/r04ah_ab.txt
