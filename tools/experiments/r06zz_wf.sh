cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06zz2_tests.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06zz2_tests.log | tail -5
rm -rf gpurun_out/profiles_r06zz_wf gpurun_out/prof_r06zz_wf_stats
bash tools/experiments/full_profile_model.sh r06zz_wf waveflow > gpurun_out/r06zz_wf.log 2>&1
head -14 profiles/r06zz_wf_kernel_stats.csv
python - <<P
import json
d=json.loads(open('gpurun_out/r06zz_wf_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['ms_per_step'],2), round(d['value']), 'box', round(d['box']['tflops_issued']), 'frac', round(r['frac'],4), 'fob', round(r.get('frac_of_box',0),3))
h=json.load(open('profiles/r06zz_wf_hbm_traffic.json'))['kernels']
print('GB/step', round(sum(v['hbm_bytes_per_launch']*v['launches'] for v in h.values())/5e9,1))
P
