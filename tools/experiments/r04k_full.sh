R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04k_gputest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04k_gputest.log
python bench.py > gpurun_out/r04k_bench.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/r04k_bench.log > gpurun_out/r04k_bench.json
python -c "
import json;d=json.load(open('gpurun_out/r04k_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('inverse_khz_16128'), d.get('inverse_khz_220672'))
c=d['cpu_baseline']; print(c['value'], c['cores'], c['all_core'], c['single_process']['samples_per_s'])
print({k:(v.get('ms_per_step'), v.get('inverse_khz_16128')) for k,v in d['other_models'].items()})
"
