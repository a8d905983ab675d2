R=$GRAFT_REPO_ROOT
cd $R
for v in nc; do
  WGFLOW_LIB=$R/variants/lib_$v.so python bench.py --model wsrglow --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v wsrglow', round(d['ms_per_step'],2), [(k['M'],k['K'],round(k['launches_per_step']),round(k['avg_us'],1)) for k in d['roofline']['kernels']['kernels'][:6]])"
done
