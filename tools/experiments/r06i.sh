# round 6: full GPU suite at the current commit, then the three bench lines (headline with everything, WaveFlow, WSRGlow)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06i_tests.txt
cat gpurun_out/r06i_tests.txt
python bench.py > gpurun_out/r06i_bench.json 2> gpurun_out/r06i_bench.err
python bench.py --model waveflow > gpurun_out/r06i_wf_bench.json 2> gpurun_out/r06i_wf_bench.err
python bench.py --model wsrglow > gpurun_out/r06i_wsr_bench.json 2> gpurun_out/r06i_wsr_bench.err
python - <<'PY'
import json
for t in ('', '_wf', '_wsr'):
    try:
        d = json.loads(open('gpurun_out/r06i%s_bench.json' % t).read().strip().splitlines()[-1])
        print(t or 'waveglow', '%.2f ms/step' % d['ms_per_step'], '%.3f M samples/s' % (d['value'] / 1e6), 'frac', round(d['roofline']['frac'], 4), 'box', d.get('box', {}).get('tflops_issued'),
              {k: round(v) for k, v in d.items() if k.startswith('inverse_khz')})
        if not t:
            print('   f32', d.get('f32_mode', {}).get('ms_per_step'), 'others', {k: v.get('ms_per_step') for k, v in d.get('other_models', {}).items() if isinstance(v, dict)})
    except Exception as e:
        print(t, 'ERR', e)
PY
