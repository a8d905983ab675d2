# round 6: the rank-2ic skip path (WG_LOWRANK) -- full GPU suite, then a same-box A/B of the training step and of the 10 s synthesis call
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06c_tests.txt
cat gpurun_out/r06c_tests.txt
for rep in 1 2; do for lrk in 0 1; do
  printf "WG_LOWRANK=%s " $lrk
  WG_LOWRANK=$lrk python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --no-box 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.2f ms/step  inv16k %.0f  inv220k %.0f  batch8 %.0f  musicnet %.0f' % (d['ms_per_step'], d['inverse_khz_16128'], d['inverse_khz_220672'], d['inverse_khz_batch8x16128'], d.get('inverse_khz_musicnet_220672', 0)))
for r in d['roofline']['kernels']['kernels'][:9]: print('   %-52s M%-5d K%-5d %6.1f us x %5.1f = %6.2f ms' % (r['kernel'][:52], r['M'], r['K'], r['avg_us'], r['launches_per_step'], r['ms_per_step']))
"
done; done 2>&1 | tee gpurun_out/r06c_ab.txt
