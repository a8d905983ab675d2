R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04v_fusemix.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inverse or infer or c2_single or full_size or reverse_mode or one_launch or graph or model_step" > gpurun_out/r04v_pytest.log 2>&1; echo "pytest rc $?" >> $O; grep -E "passed|failed" gpurun_out/r04v_pytest.log | tail -2 >> $O
for rep in 1 2 3; do
python tools/experiments/infer_latency.py 63 2>&1 | grep "single call" >> $O
done
python tools/experiments/infer_latency.py 862 2>&1 | grep "single call" >> $O
cat $O
