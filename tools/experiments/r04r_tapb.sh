R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04r_tapb.txt
: > $O
for v in t0 t3; do
  WGFLOW_LIB=$R/variants/lib_$v.so python bench.py --model waveflow --no-inverse --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v waveflow', round(d['ms_per_step'],2), [(k['M'],k['K'],round(k['avg_us'],1)) for k in d['roofline']['kernels']['kernels'][:4]])" >> $O
  WGFLOW_LIB=$R/variants/lib_$v.so python bench.py --no-cpu --no-extra --no-inverse --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v waveglow', round(d['ms_per_step'],2), [(k['M'],k['K'],round(k['avg_us'],1)) for k in d['roofline']['kernels']['kernels'][:6]])" >> $O
done
cat $O
