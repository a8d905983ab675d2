cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -s -k "waveflow or wn2d or layer2d" > gpurun_out/r06r_tests.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06r_tests.log | tail -3; grep -E "WaveFlow 12 x 16000" gpurun_out/r06r_tests.log
for rep in 1 2; do for lrk in 0 1; do
  printf "waveflow WG_LOWRANK=%s " $lrk
  WG_LOWRANK=$lrk python bench.py --model waveflow --steps 10 --warmup 3 --no-box --no-inverse 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.2f ms/step' % d['ms_per_step'])
for r in d['roofline']['kernels']['kernels'][:9]: print('   %-52s M%-7d K%-5d %6.1f us x %5.1f = %6.2f ms' % (r['kernel'][:52], r['M'], r['K'], r['avg_us'], r['launches_per_step'], r['ms_per_step']))
"
done; done 2>&1 | tee gpurun_out/r06r_wf_ab.txt
