cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06zz_bench.json 2> gpurun_out/r06zz_bench.err
python tools/profile_summary.py --shapes r06zz gpurun_out/r06zz_bench.json; cp profiles/r06zz_shape_rooflines.csv gpurun_out/
python - <<P
import json
d=json.loads(open('gpurun_out/r06zz_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['ms_per_step'],2), round(d['value']), 'box', round(d['box']['tflops_issued']), 'frac', round(r['frac'],4), 'fob', round(r.get('frac_of_box',0),3), 'gate alone', r['gate_conv_alone']['launch_ms'], r['gate_conv_alone']['frac'])
print({k:round(v) for k,v in d.items() if k.startswith('inverse_khz')})
print(d['f32_mode']['ms_per_step'], {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('other_models',{}).items()})
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
P
