R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04h_bisect.txt
: > $O
for v in q0 q1 q2 q4 q8 q15; do
  WGFLOW_LIB=$R/variants/lib_$v.so timeout 300 python bench.py --no-cpu --no-extra --no-inverse --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), [ (k['kernel'][:12],k['M'],k['K'],round(k['launches_per_step']),round(k['avg_us'],1)) for k in d['roofline']['kernels']['kernels'][:8] if k['kernel'][:1] in '5g'])" >> $O
done
cat $O
