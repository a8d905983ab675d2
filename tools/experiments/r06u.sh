cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 600 python bench.py --gpus 2 --oversubscribe --steps 3 --warmup 1 > gpurun_out/r06u_2rank.json 2> gpurun_out/r06u_2rank.err; echo rc=$?
tail -c 1500 gpurun_out/r06u_2rank.json | cut -c1-1500
tail -3 gpurun_out/r06u_2rank.err | cut -c1-300
timeout 900 python bench.py > gpurun_out/r06u_bench.json 2> gpurun_out/r06u_bench.err; echo rc=$?
python -c "
import json
d=json.loads(open('gpurun_out/r06u_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['box']['tflops_issued'], d['roofline']['frac'], d['roofline']['frac_of_box'])
print({k:v for k,v in d.items() if k.startswith('inverse_khz')})
print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('other_models',{}).items()})
"
