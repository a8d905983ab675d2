R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04l_wf2.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "waveflow or wf or graph" > gpurun_out/r04l_pytest.log 2>&1; echo "pytest rc $?" >> $O; tail -3 gpurun_out/r04l_pytest.log >> $O
bash tools/experiments/r04i_wf.sh >> $O 2>&1
python tools/experiments/wf_infer_profile.py 16128 2 2>&1 | tail -1 >> $O
python tools/experiments/wf_infer_profile.py 220672 1 2>&1 | tail -1 >> $O
python tools/experiments/infer_latency.py 63 2>&1 | grep "single call" >> $O
cat $O
