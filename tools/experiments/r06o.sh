cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "inverse or infer or wide_batch or shape_sweep or flattened or full_size_properties or trainer_step or waveflow_model or model_step_vs_oracle or sixteen_rows or one_launch_layer or layer_launch_of" > gpurun_out/r06o_tests.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06o_tests.log | tail -3
python bench.py --no-cpu > gpurun_out/r06o_bench.json 2> gpurun_out/r06o_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06o_bench.json').read().strip().splitlines()[-1])
print('%.2f ms/step' % d['ms_per_step'], d['box']['tflops_issued'], {k:round(v) for k,v in d.items() if k.startswith('inverse_khz')})
k=d['roofline']['kernels']; print(k['timed_ms_per_step'], [(r['kernel'][:20], r['M'], r['K'], round(r['avg_us'],1), r['launches_per_step']) for r in k['kernels'][:5]])
print({c: v for c, v in d['inverse_roofline']['cases'].items()})
PY
