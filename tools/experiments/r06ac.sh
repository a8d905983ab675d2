cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in variants/lib_ncr8.so constant-memory-waveglow_amd/csrc/libwgflow.so; do
WGFLOW_LIB=$GRAFT_REPO_ROOT/$v python bench.py --model waveflow --steps 10 --warmup 3 > gpurun_out/r06ac_wf.json 2> gpurun_out/r06ac_wf.err
python - <<P
import json
d=json.loads(open('gpurun_out/r06ac_wf.json').read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step'],2), 'ms  box', round(d['box']['tflops_issued']))
P
done
done
