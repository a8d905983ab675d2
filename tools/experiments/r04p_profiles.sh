# round 4 profiles of record: the headline command (bench line, kernel stats, HBM traffic, PMC, per-shape rooflines), WaveFlow and WSRGlow
R=$GRAFT_REPO_ROOT
cd $R
bash tools/experiments/full_profile.sh r04p > gpurun_out/r04p_full.log 2>&1
st=$(ls -t $(find gpurun_out/prof_r04p_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_r04p_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_r04p_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py r04p $st $fe $wr
python tools/profile_summary.py --pmc r04p $(find gpurun_out/prof_r04p_mfma gpurun_out/prof_r04p_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes r04p gpurun_out/r04p_bench.json
mkdir -p gpurun_out/profiles_r04p && cp profiles/r04p_* gpurun_out/profiles_r04p/ && cp gpurun_out/r04p_bench.json gpurun_out/profiles_r04p/
rm -rf gpurun_out/prof_r04p_fetch gpurun_out/prof_r04p_write gpurun_out/prof_r04p_mfma gpurun_out/prof_r04p_lds
bash tools/experiments/full_profile_model.sh r04_wf waveflow > gpurun_out/r04p_wf.log 2>&1
bash tools/experiments/full_profile_model.sh r04_wsr wsrglow > gpurun_out/r04p_wsr.log 2>&1
head -12 profiles/r04p_kernel_stats.csv; head -12 profiles/r04_wf_kernel_stats.csv; head -8 profiles/r04_wsr_kernel_stats.csv
python -c "
import json
for t in ('r04p','r04_wf','r04_wsr'):
    d=json.load(open('profiles/%s_hbm_traffic.json'%t))['kernels']
    print(t, {k[:34]:round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if v['hbm_bytes_per_launch']>5e7})
"
