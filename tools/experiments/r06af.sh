cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in constant-memory-waveglow_amd/csrc/libwgflow.so variants/lib_sync4.so variants/lib_sync16.so; do
WGFLOW_LIB=$GRAFT_REPO_ROOT/$v python bench.py --model waveflow --steps 10 --warmup 3 > gpurun_out/r06af_wf.json 2> gpurun_out/r06af_wf.err
python - <<P
import json
d=json.loads(open('gpurun_out/r06af_wf.json').read().strip().splitlines()[-1])
w=[k for k in d['roofline']['kernels']['kernels'] if k['kernel'].startswith('weight gradient') and k['M']>0]
print('$v', round(d['ms_per_step'],2), 'ms  box', round(d['box']['tflops_issued']), 'wgrad', [round(k['avg_us'],1) for k in w], 'loss', d.get('loss'))
P
done
done
