# the one-launch layer, clean A/B (nothing else running on the host): synthesis latency and kernel timeline
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04d_ab.txt
: > $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch_layer" > gpurun_out/r04d_pytest.log 2>&1; echo "pytest rc $?" >> $O; tail -3 gpurun_out/r04d_pytest.log >> $O
for rep in 1 2; do for f in 1 0; do
  echo "WG_LAYER_FUSION=$f" >> $O
  WG_LAYER_FUSION=$f python tools/experiments/infer_latency.py 63 2>&1 | grep "single call" >> $O
  WG_LAYER_FUSION=$f python tools/experiments/infer_latency.py 862 2>&1 | grep "single call" >> $O
  WG_LAYER_FUSION=$f python tools/experiments/wf_infer_profile.py 16128 2 2>&1 | tail -1 >> $O
done; done
WG_LAYER_FUSION=1 python tools/experiments/wf_infer_profile.py 220672 1 2>&1 | tail -1 >> $O
WG_LAYER_FUSION=0 python tools/experiments/wf_infer_profile.py 220672 1 2>&1 | tail -1 >> $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04d_infer -- python3 $R/tools/experiments/infer_profile.py 63 > $R/gpurun_out/r04d_infer.log 2>&1
python3 $R/tools/experiments/trace_gaps.py $R/gpurun_out/prof_r04d_infer 0.5 12 >> $O
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04d_wfinv -- python3 $R/tools/experiments/wf_infer_profile.py 16128 1 > $R/gpurun_out/r04d_wfinv.log 2>&1
python3 $R/tools/experiments/trace_gaps.py $R/gpurun_out/prof_r04d_wfinv 0.5 12 >> $O
rm -rf $R/gpurun_out/prof_r04d_wfinv $R/gpurun_out/prof_r04d_infer
cat $O
