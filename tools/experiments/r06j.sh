cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06j_tests.txt
cat gpurun_out/r06j_tests.txt
for rep in 1 2; do for lrk in 0 1; do
  printf "wsrglow WG_LOWRANK=%s " $lrk
  WG_LOWRANK=$lrk python bench.py --model wsrglow --steps 10 --warmup 3 --no-box 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.2f ms/step' % d['ms_per_step'])
for r in d['roofline']['kernels']['kernels'][:8]: print('   %-52s M%-5d K%-5d %6.1f us x %5.1f = %6.2f ms' % (r['kernel'][:52], r['M'], r['K'], r['avg_us'], r['launches_per_step'], r['ms_per_step']))
"
done; done 2>&1 | tee gpurun_out/r06j_wsr_ab.txt
