R=$GRAFT_REPO_ROOT
cd $R
python bench.py --model waveflow --no-inverse --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r04i_wf_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04i_wf_bench.json'))
print(d['ms_per_step'])
k=d['roofline']['kernels']
print('timed', k['timed_ms_per_step'])
for r in k['kernels']:
    print('  %-50s M%8d K%5d cols %7d n/step %5.1f avg %7.1f us  ms/step %6.2f  TF %6.1f GB/s %6.0f %s %.2f'%(r['kernel'][:50],r['M'],r['K'],r['columns'],r['launches_per_step'],r['avg_us'],r['ms_per_step'],r['tflops_algorithmic'],r['gbs_algorithmic'],r['bound'],r['frac']))
PY
