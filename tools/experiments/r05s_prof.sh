cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05s -- python3 $R/tools/experiments/infer_profile.py 63 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_r05s/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s %6s %9.2f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
