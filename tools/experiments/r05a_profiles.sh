# round 5 profiles, headline command (bench line, kernel stats, HBM traffic, PMC, per-shape rooflines)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/experiments/full_profile.sh r05a > gpurun_out/r05a_full.log 2>&1
st=$(ls -t $(find gpurun_out/prof_r05a_stats -name "*kernel_stats.csv") | head -1)
fe=$(ls -t $(find gpurun_out/prof_r05a_fetch -name "*counter_collection.csv") | head -1)
wr=$(ls -t $(find gpurun_out/prof_r05a_write -name "*counter_collection.csv") | head -1)
python tools/profile_summary.py r05a $st $fe $wr
python tools/profile_summary.py --pmc r05a $(find gpurun_out/prof_r05a_mfma gpurun_out/prof_r05a_lds -name "*counter_collection.csv")
python tools/profile_summary.py --shapes r05a gpurun_out/r05a_bench.json
mkdir -p gpurun_out/profiles_r05a && cp profiles/r05a_* gpurun_out/profiles_r05a/ && cp gpurun_out/r05a_bench.json gpurun_out/profiles_r05a/
rm -rf gpurun_out/prof_r05a_fetch gpurun_out/prof_r05a_write gpurun_out/prof_r05a_mfma gpurun_out/prof_r05a_lds
python -c "
import json
for t in ('r05a',):
    d=json.load(open('profiles/%s_hbm_traffic.json'%t))['kernels']
    print(t, {k[:34]:round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items() if v['hbm_bytes_per_launch']>5e7})
"
