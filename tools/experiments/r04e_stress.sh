R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04e_stress.txt
: > $O
for i in 1 2 3 4 5 6; do for n in micro c1; do
  WGFLOW_LIB=$R/variants/lib_lv0.so timeout 120 python tools/experiments/layer_stress.py $n 4 cold 2>&1 | grep "fused repetitions" >> $O
  timeout 120 python tools/experiments/layer_stress.py $n 4 cold 2>&1 | grep "fused repetitions" >> $O
done; done
timeout 120 python tools/experiments/layer_stress.py c2 4 cold 2>&1 | grep "fused repetitions" >> $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch_layer" > gpurun_out/r04e_pytest.log 2>&1; echo "pytest rc $?" >> $O; tail -3 gpurun_out/r04e_pytest.log >> $O
for rep in 1 2; do for f in 1 0; do
  echo "WG_LAYER_FUSION=$f" >> $O
  WG_LAYER_FUSION=$f timeout 120 python tools/experiments/infer_latency.py 63 2>&1 | grep "single call" >> $O
  WG_LAYER_FUSION=$f timeout 200 python tools/experiments/wf_infer_profile.py 16128 2 2>&1 | tail -1 >> $O
done; done
cat $O
