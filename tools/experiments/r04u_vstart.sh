R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04u_vstart.txt
: > $O
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04u_gputest.log 2>&1; echo "pytest rc $?" >> $O; grep -E "passed|failed" gpurun_out/r04u_gputest.log | tail -2 >> $O
for rep in 1 2; do
python tools/experiments/infer_latency.py 63 2>&1 | grep "single call" >> $O
WGFLOW_LIB=$R/variants/lib_t0.so python tools/experiments/infer_latency.py 63 2>&1 | grep "single call" >> $O
done
python bench.py --no-cpu --no-extra --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d.get('inverse_khz_16128'), d.get('inverse_khz_220672'))" >> $O
WGFLOW_LIB=$R/variants/lib_t0.so python bench.py --no-cpu --no-extra --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('old', d['ms_per_step'], d.get('inverse_khz_16128'), d.get('inverse_khz_220672'))" >> $O
cat $O
