cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "c1 or c2 or sweep or micro or inverse or trainer" > gpurun_out/r06v_tests.log 2>&1
grep -E "passed|failed|FAILED|^E  " gpurun_out/r06v_tests.log | head -12
for sw in 0 1 0 1; do
  WG_START_FOLD=$sw python bench.py --steps 10 --warmup 3 > gpurun_out/r06v_b$sw.json 2> gpurun_out/r06v_b$sw.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r06v_b$sw.json').read().strip().splitlines()[-1])
print('WG_START_FOLD=$sw', round(d['ms_per_step'],2), 'ms  box', round(d['box']['tflops_issued']), ' inv10s', round(d['inverse_khz_220672']), ' musicnet', round(d['inverse_khz_musicnet_220672']), 'loss', d.get('loss'))
for k in d['roofline']['kernels']['kernels'][:9]:
    print("   %-58s M%-8d K%-6d %7.1f us x %5.1f = %6.2f ms"%(k['kernel'][:58],k['M'],k['K'],k['avg_us'],k['launches_per_step'],k['ms_per_step']))
P
done
