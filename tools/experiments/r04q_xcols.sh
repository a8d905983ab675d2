R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04q_xcols.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -m gpu -k "wsr or inverse or infer or c2_single or full_size or wide_batch" > gpurun_out/r04q_pytest.log 2>&1; echo "pytest rc $?" >> $O; tail -3 gpurun_out/r04q_pytest.log >> $O
for v in 1 0; do
  if [ $v = 0 ]; then export WGFLOW_LIB=$R/variants/lib_a0.so; else unset WGFLOW_LIB; fi
  python bench.py --model wsrglow --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('xcols=$v wsrglow', round(d['ms_per_step'],2), [(k['M'],k['K'],round(k['avg_us'],1)) for k in d['roofline']['kernels']['kernels'][:6]])" >> $O
  python tools/experiments/infer_latency.py 862 2>&1 | grep "single call" >> $O
done
cat $O
