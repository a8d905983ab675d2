# round 6 profiles of record (state: the rank-2ic skip path with the gate convs writing their share of out): the headline command (bench line, kernel stats, HBM traffic, PMC, per-shape rooflines), WaveFlow and WSRGlow
R=$GRAFT_REPO_ROOT
cd $R
bash tools/experiments/full_profile.sh r06y > gpurun_out/r06y_full.log 2>&1
st=$(ls -t $(find gpurun_out/prof_r06y_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_r06y_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_r06y_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py r06y $st $fe $wr
python tools/profile_summary.py --pmc r06y $(find gpurun_out/prof_r06y_mfma gpurun_out/prof_r06y_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes r06y gpurun_out/r06y_bench.json
mkdir -p gpurun_out/profiles_r06y && cp profiles/r06y_* gpurun_out/profiles_r06y/ && cp gpurun_out/r06y_bench.json gpurun_out/profiles_r06y/
rm -rf gpurun_out/prof_r06y_fetch gpurun_out/prof_r06y_write gpurun_out/prof_r06y_mfma gpurun_out/prof_r06y_lds
bash tools/experiments/full_profile_model.sh r06y_wf waveflow > gpurun_out/r06y_wf.log 2>&1
bash tools/experiments/full_profile_model.sh r06y_wsr wsrglow > gpurun_out/r06y_wsr.log 2>&1
head -12 profiles/r06y_kernel_stats.csv; head -12 profiles/r06y_wf_kernel_stats.csv; head -8 profiles/r06y_wsr_kernel_stats.csv
python -c "
import json
for t in ('r06y','r06y_wf','r06y_wsr'):
    d=json.load(open('profiles/%s_hbm_traffic.json'%t))['kernels']
    print(t, {k[:34]:round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if v['hbm_bytes_per_launch']>5e7})
"
