# round 4: the headline-shape oracle test, the --model bench legs (1 rank, launcher 1 rank, 2 gloo ranks on the one GPU), full default line
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c2_full_batch or (weight_gradient_kernel_plans and 24-2000)" -s > gpurun_out/r04b_pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04b_pytest.log
for m in waveflow wsrglow; do
  python bench.py --model $m --steps 5 --warmup 2 > gpurun_out/r04b_bench_$m.log 2>&1; echo "$m rc $?"; tail -1 gpurun_out/r04b_bench_$m.log > gpurun_out/r04b_bench_$m.json; cut -c1-700 gpurun_out/r04b_bench_$m.json
  python bench.py --model $m --gpus 1 --spawn --steps 5 --warmup 2 > gpurun_out/r04b_bench_${m}_1rank.log 2>&1; echo "$m spawn rc $?"; tail -1 gpurun_out/r04b_bench_${m}_1rank.log > gpurun_out/r04b_bench_${m}_1rank.json; python -c "import json;d=json.load(open('gpurun_out/r04b_bench_${m}_1rank.json'));print(d['ms_per_step'], d.get('comm'))"
  python bench.py --model $m --gpus 2 --oversubscribe --steps 3 --warmup 1 > gpurun_out/r04b_bench_${m}_2ranks.log 2>&1; echo "$m 2ranks rc $?"; tail -1 gpurun_out/r04b_bench_${m}_2ranks.log > gpurun_out/r04b_bench_${m}_2ranks.json; python -c "import json;d=json.load(open('gpurun_out/r04b_bench_${m}_2ranks.json'));print(d['n_gpus'], d['ms_per_step'], d.get('comm'))"
done
python bench.py --gpus 1 --spawn --steps 5 --warmup 2 --no-cpu --no-extra --no-inverse > gpurun_out/r04b_bench_wg_1rank.log 2>&1; tail -1 gpurun_out/r04b_bench_wg_1rank.log | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d.get('comm'))"
python bench.py > gpurun_out/r04b_bench.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/r04b_bench.log > gpurun_out/r04b_bench.json
python -c "
import json;d=json.load(open('gpurun_out/r04b_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('inverse_khz_16128'), d.get('inverse_khz_220672'))
c=d['cpu_baseline']; print(c['value'], c['cores'], c['all_core'], c['single_process'], c['physical_cores'])
print({k:(v.get('ms_per_step'), v.get('inverse_khz_16128')) for k,v in d['other_models'].items()})
print(d['f32_mode']['ms_per_step'], d['f32_mode']['steps'])
"
