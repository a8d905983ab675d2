# round 5 profiles of record (final commit of the round): the headline command (bench line, kernel stats, HBM traffic, PMC, per-shape rooflines), WaveFlow and WSRGlow
R=$GRAFT_REPO_ROOT
cd $R
bash tools/experiments/full_profile.sh r05z > gpurun_out/r05z_full.log 2>&1
st=$(ls -t $(find gpurun_out/prof_r05z_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_r05z_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_r05z_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py r05z $st $fe $wr
python tools/profile_summary.py --pmc r05z $(find gpurun_out/prof_r05z_mfma gpurun_out/prof_r05z_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes r05z gpurun_out/r05z_bench.json
mkdir -p gpurun_out/profiles_r05z && cp profiles/r05z_* gpurun_out/profiles_r05z/ && cp gpurun_out/r05z_bench.json gpurun_out/profiles_r05z/
rm -rf gpurun_out/prof_r05z_fetch gpurun_out/prof_r05z_write gpurun_out/prof_r05z_mfma gpurun_out/prof_r05z_lds
bash tools/experiments/full_profile_model.sh r05z_wf waveflow > gpurun_out/r05z_wf.log 2>&1
bash tools/experiments/full_profile_model.sh r05z_wsr wsrglow > gpurun_out/r05z_wsr.log 2>&1
head -12 profiles/r05z_kernel_stats.csv; head -12 profiles/r05z_wf_kernel_stats.csv; head -8 profiles/r05z_wsr_kernel_stats.csv
python -c "
import json
for t in ('r05z','r05z_wf','r05z_wsr'):
    d=json.load(open('profiles/%s_hbm_traffic.json'%t))['kernels']
    print(t, {k[:34]:round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if v['hbm_bytes_per_launch']>5e7})
"
