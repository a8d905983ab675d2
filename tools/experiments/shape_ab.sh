# usage: bash tools/experiments/shape_ab.sh v1 v2 ...: per-shape conv / weight-gradient durations (bench.py's roofline.kernels) per variant library
for v in "$@"; do
  echo "== $v"
  WGFLOW_LIB=variants/lib_$v.so python bench.py --steps 3 --warmup 1 --no-cpu --no-extra --no-inverse 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  step %.2f ms'%d['ms_per_step'])
for k in d['roofline']['kernels']['kernels']:
    print('  %-44s M%-4d K%-5d %6.1f us x %5.1f/step = %5.2f ms   %.2f TB/s'%(k['kernel'][:44],k['M'],k['K'],k['avg_us'],k['launches_per_step'],k['ms_per_step'],k['gbs_algorithmic']/1e3))
"
done
