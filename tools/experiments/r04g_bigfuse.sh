R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04g_bigfuse.txt
: > $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "layer_launch_of_the_training_shapes" > gpurun_out/r04g_pytest.log 2>&1; echo "pytest rc $?" >> $O; tail -5 gpurun_out/r04g_pytest.log >> $O
for rep in 1 2; do for f in 1 0; do
  WG_LAYER_FUSION_BIG=$f timeout 300 python bench.py --no-cpu --no-extra --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f', d['ms_per_step'], d['value'], d.get('inverse_khz_16128'), d.get('inverse_khz_220672'), [ (k['kernel'][:28],k['M'],k['K'],round(k['launches_per_step']),round(k['avg_us'],1)) for k in d['roofline']['kernels']['kernels'][:8]])" >> $O
done; done
cat $O
