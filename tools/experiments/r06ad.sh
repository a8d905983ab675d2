cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "waveflow or wn2d or layer2d" > gpurun_out/r06ad_tests.log 2>&1
grep -E "passed|failed|FAILED|^E  " gpurun_out/r06ad_tests.log | head -8
for i in 1 2; do
for v in variants/lib_so1d.so constant-memory-waveglow_amd/csrc/libwgflow.so; do
WGFLOW_LIB=$GRAFT_REPO_ROOT/$v python bench.py --model waveflow --steps 10 --warmup 3 > gpurun_out/r06ad_wf.json 2> gpurun_out/r06ad_wf.err
python - <<P
import json
d=json.loads(open('gpurun_out/r06ad_wf.json').read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step'],2), 'ms  box', round(d['box']['tflops_issued']), 'loss', d.get('loss'))
for k in d['roofline']['kernels']['kernels'][:8]:
    print("   %-58s M%-8d K%-6d %7.1f us x %5.1f = %6.2f ms"%(k['kernel'][:58],k['M'],k['K'],k['avg_us'],k['launches_per_step'],k['ms_per_step']))
P
done
done
