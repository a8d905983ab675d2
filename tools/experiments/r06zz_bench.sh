cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r06zz_bench.json 2> gpurun_out/r06zz_bench.err
python bench.py --model waveflow > gpurun_out/r06zz_wf_bench.json 2> gpurun_out/r06zz_wf_bench.err
python bench.py --model wsrglow > gpurun_out/r06zz_wsr_bench.json 2> gpurun_out/r06zz_wsr_bench.err
for t in r06zz r06zz_wf r06zz_wsr; do python tools/profile_summary.py --shapes $t gpurun_out/${t}_bench.json; cp profiles/${t}_shape_rooflines.csv gpurun_out/; done
python - <<P
import json
for t in ('r06zz','r06zz_wf','r06zz_wsr'):
    d=json.loads(open('gpurun_out/%s_bench.json'%t).read().strip().splitlines()[-1]); r=d['roofline']
    print(t, round(d['ms_per_step'],2), round(d['value']), 'box', round(d['box']['tflops_issued']), 'frac', round(r['frac'],4), 'fob', round(r.get('frac_of_box',0),3), 'launch_ms', round(r['launch_ms'],4), r.get('gate_conv_alone',{}).get('launch_ms'), r.get('gate_conv_alone',{}).get('frac'))
P
