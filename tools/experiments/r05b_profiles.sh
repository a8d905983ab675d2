# round 5 profiles of record: the headline command (bench line, kernel stats, HBM traffic, PMC, per-shape rooflines), WaveFlow and WSRGlow
R=$GRAFT_REPO_ROOT
cd $R
bash tools/experiments/full_profile.sh r05b > gpurun_out/r05b_full.log 2>&1
st=$(ls -t $(find gpurun_out/prof_r05b_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_r05b_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_r05b_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py r05b $st $fe $wr
python tools/profile_summary.py --pmc r05b $(find gpurun_out/prof_r05b_mfma gpurun_out/prof_r05b_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes r05b gpurun_out/r05b_bench.json
mkdir -p gpurun_out/profiles_r05b && cp profiles/r05b_* gpurun_out/profiles_r05b/ && cp gpurun_out/r05b_bench.json gpurun_out/profiles_r05b/
rm -rf gpurun_out/prof_r05b_fetch gpurun_out/prof_r05b_write gpurun_out/prof_r05b_mfma gpurun_out/prof_r05b_lds
bash tools/experiments/full_profile_model.sh r05b_wf waveflow > gpurun_out/r05b_wf.log 2>&1
bash tools/experiments/full_profile_model.sh r05b_wsr wsrglow > gpurun_out/r05b_wsr.log 2>&1
head -12 profiles/r05b_kernel_stats.csv; head -12 profiles/r05b_wf_kernel_stats.csv; head -8 profiles/r05b_wsr_kernel_stats.csv
python -c "
import json
for t in ('r05b','r05b_wf','r05b_wsr'):
    d=json.load(open('profiles/%s_hbm_traffic.json'%t))['kernels']
    print(t, {k[:34]:round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if v['hbm_bytes_per_launch']>5e7})
"
