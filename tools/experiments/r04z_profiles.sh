# round 4 profiles of record: the headline command (bench line, kernel stats, HBM traffic, PMC, per-shape rooflines), WaveFlow and WSRGlow
R=$GRAFT_REPO_ROOT
cd $R
bash tools/experiments/full_profile.sh r04z > gpurun_out/r04z_full.log 2>&1
st=$(ls -t $(find gpurun_out/prof_r04z_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_r04z_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_r04z_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py r04z $st $fe $wr
python tools/profile_summary.py --pmc r04z $(find gpurun_out/prof_r04z_mfma gpurun_out/prof_r04z_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes r04z gpurun_out/r04z_bench.json
mkdir -p gpurun_out/profiles_r04z && cp profiles/r04z_* gpurun_out/profiles_r04z/ && cp gpurun_out/r04z_bench.json gpurun_out/profiles_r04z/
rm -rf gpurun_out/prof_r04z_fetch gpurun_out/prof_r04z_write gpurun_out/prof_r04z_mfma gpurun_out/prof_r04z_lds
bash tools/experiments/full_profile_model.sh r04z_wf waveflow > gpurun_out/r04z_wf.log 2>&1
bash tools/experiments/full_profile_model.sh r04z_wsr wsrglow > gpurun_out/r04z_wsr.log 2>&1
head -12 profiles/r04z_kernel_stats.csv; head -12 profiles/r04z_wf_kernel_stats.csv; head -8 profiles/r04z_wsr_kernel_stats.csv
python -c "
import json
for t in ('r04z','r04z_wf','r04z_wsr'):
    d=json.load(open('profiles/%s_hbm_traffic.json'%t))['kernels']
    print(t, {k[:34]:round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if v['hbm_bytes_per_launch']>5e7})
"
