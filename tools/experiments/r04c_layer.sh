# round 4: the one-launch layer: parity (bit-equality with two launches, model tests), synthesis timing A/B, CPU scaling probe
R=$GRAFT_REPO_ROOT
cd $R
python tools/experiments/cpu_scaling_probe.py > gpurun_out/r04c_cpu_probe.log 2>&1 &
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch_layer or model_inverse_and_infer or c2_single_segment or full_size_properties or wf64 or wf8 or waveflow" > gpurun_out/r04c_pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04c_pytest.log
for f in 1 0; do
  WG_LAYER_FUSION=$f python tools/experiments/infer_latency.py 63 2>&1 | grep "single call"
  WG_LAYER_FUSION=$f python tools/experiments/infer_latency.py 862 2>&1 | grep "single call"
  WG_LAYER_FUSION=$f python tools/experiments/wf_infer_profile.py 16128 2 2>&1 | tail -1
  WG_LAYER_FUSION=$f python tools/experiments/wf_infer_profile.py 220672 1 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04c_infer -- python3 $R/tools/experiments/infer_profile.py 63 > $R/gpurun_out/r04c_infer.log 2>&1
python3 $R/tools/experiments/trace_gaps.py $R/gpurun_out/prof_r04c_infer 0.5 12
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r04c_wfinv -- python3 $R/tools/experiments/wf_infer_profile.py 16128 1 > $R/gpurun_out/r04c_wfinv.log 2>&1
python3 $R/tools/experiments/trace_gaps.py $R/gpurun_out/prof_r04c_wfinv 0.5 12
rm -rf $R/gpurun_out/prof_r04c_wfinv
wait
cat $R/gpurun_out/r04c_cpu_probe.log
