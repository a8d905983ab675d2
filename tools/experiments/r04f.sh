R=$GRAFT_REPO_ROOT
cd $R
python tools/experiments/cpu_scaling_probe.py 2>&1 | grep -v "amdgpu.ids\|Warn\|WeightNorm" > gpurun_out/r04f_cpu_probe.log
grep MemTotal /proc/meminfo >> gpurun_out/r04f_cpu_probe.log
cat gpurun_out/r04f_cpu_probe.log
bash tools/experiments/full_profile_model.sh r04_wf waveflow
bash tools/experiments/full_profile_model.sh r04_wsr wsrglow
for g in 0 1; do WG_GRAPHS=$g python tools/experiments/infer_latency.py 63 2>&1 | grep "single call"; done
