cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "waveflow or wn2d" > gpurun_out/r06ag_tests.log 2>&1
grep -E "passed|failed|FAILED|^E  " gpurun_out/r06ag_tests.log | head -8
for i in 1 2; do
for v in variants/lib_prev.so constant-memory-waveglow_amd/csrc/libwgflow.so; do
WGFLOW_LIB=$GRAFT_REPO_ROOT/$v python bench.py --model waveflow --steps 10 --warmup 3 > gpurun_out/r06ag_wf.json 2> gpurun_out/r06ag_wf.err
python - <<P
import json
d=json.loads(open('gpurun_out/r06ag_wf.json').read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step'],2), 'ms  box', round(d['box']['tflops_issued']), 'loss', d.get('loss'))
P
done
done
