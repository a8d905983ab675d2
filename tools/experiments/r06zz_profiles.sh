# round 6 profiles of record at the last kernel commit (rank-2ic skip path, gate convs writing their share of out, tanh not kept): headline, WaveFlow, WSRGlow
R=$GRAFT_REPO_ROOT
cd $R
bash tools/experiments/full_profile.sh r06zz > gpurun_out/r06zz_full.log 2>&1
st=$(ls -t $(find gpurun_out/prof_r06zz_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_r06zz_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_r06zz_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py r06zz $st $fe $wr
python tools/profile_summary.py --pmc r06zz $(find gpurun_out/prof_r06zz_mfma gpurun_out/prof_r06zz_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes r06zz gpurun_out/r06zz_bench.json
mkdir -p gpurun_out/profiles_r06zz && cp profiles/r06zz_[bhkps]* gpurun_out/profiles_r06zz/ && cp gpurun_out/r06zz_bench.json gpurun_out/profiles_r06zz/
rm -rf gpurun_out/prof_r06zz_fetch gpurun_out/prof_r06zz_write gpurun_out/prof_r06zz_mfma gpurun_out/prof_r06zz_lds
bash tools/experiments/full_profile_model.sh r06zz_wf waveflow > gpurun_out/r06zz_wf.log 2>&1
bash tools/experiments/full_profile_model.sh r06zz_wsr wsrglow > gpurun_out/r06zz_wsr.log 2>&1
head -12 profiles/r06zz_kernel_stats.csv; head -12 profiles/r06zz_wf_kernel_stats.csv; head -8 profiles/r06zz_wsr_kernel_stats.csv
python -c "
import json
for t in ('r06zz','r06zz_wf','r06zz_wsr'):
    d=json.load(open('profiles/%s_hbm_traffic.json'%t))['kernels']
    print(t, {k[:34]:round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if v['hbm_bytes_per_launch']>5e7})
"
