cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "waveflow or wn2d or layer2d or wsrglow" > gpurun_out/r06x_tests.log 2>&1
grep -E "passed|failed|FAILED|^E  " gpurun_out/r06x_tests.log | head -12
for sw in 0 1 0 1; do
  for m in waveflow wsrglow; do
  WG_START_FOLD=$sw python bench.py --model $m --steps 10 --warmup 3 > gpurun_out/r06x_${m}_$sw.json 2> gpurun_out/r06x_${m}_$sw.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r06x_${m}_$sw.json').read().strip().splitlines()[-1])
print('$m WG_START_FOLD=$sw', round(d['ms_per_step'],2), 'ms  box', round(d['box']['tflops_issued']), 'loss', d.get('loss'))
P
  done
done
