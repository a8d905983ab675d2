cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c2_full or plans or (golden and (c1 or micro))" > gpurun_out/r04ap_test.txt 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r04ap_test.txt
bash tools/experiments/ab_bench.sh base w12 > gpurun_out/r04ap_ab.txt 2>&1; cat gpurun_out/r04ap_ab.txt
