R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04n_cg2.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "waveflow or wf" > gpurun_out/r04n_pytest.log 2>&1; echo "pytest rc $?" >> $O; tail -3 gpurun_out/r04n_pytest.log >> $O
bash tools/experiments/r04i_wf.sh >> $O 2>&1
cat $O
