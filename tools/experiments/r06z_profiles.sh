# round 6 profiles of record at the last kernel commit (rank-2ic skip path, gate convs writing their share of out, tanh not kept): headline, WaveFlow, WSRGlow
R=$GRAFT_REPO_ROOT
cd $R
bash tools/experiments/full_profile.sh r06z > gpurun_out/r06z_full.log 2>&1
st=$(ls -t $(find gpurun_out/prof_r06z_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_r06z_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_r06z_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py r06z $st $fe $wr
python tools/profile_summary.py --pmc r06z $(find gpurun_out/prof_r06z_mfma gpurun_out/prof_r06z_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes r06z gpurun_out/r06z_bench.json
mkdir -p gpurun_out/profiles_r06z && cp profiles/r06z_* gpurun_out/profiles_r06z/ && cp gpurun_out/r06z_bench.json gpurun_out/profiles_r06z/
rm -rf gpurun_out/prof_r06z_fetch gpurun_out/prof_r06z_write gpurun_out/prof_r06z_mfma gpurun_out/prof_r06z_lds
bash tools/experiments/full_profile_model.sh r06z_wf waveflow > gpurun_out/r06z_wf.log 2>&1
bash tools/experiments/full_profile_model.sh r06z_wsr wsrglow > gpurun_out/r06z_wsr.log 2>&1
head -12 profiles/r06z_kernel_stats.csv; head -12 profiles/r06z_wf_kernel_stats.csv; head -8 profiles/r06z_wsr_kernel_stats.csv
python -c "
import json
for t in ('r06z','r06z_wf','r06z_wsr'):
    d=json.load(open('profiles/%s_hbm_traffic.json'%t))['kernels']
    print(t, {k[:34]:round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if v['hbm_bytes_per_launch']>5e7})
"
