cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export WGFLOW_LIB=$R/variants/lib_seamdbg.so
for sw in 1 2 3; do
export WG_INV_SEAM=$sw
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05s_$sw -- python3 $R/tools/experiments/infer_profile.py 63 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob('$R/gpurun_out/prof_r05s_$sw/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f))):
    if 'end_affine' in r['Name']: print("$sw %-60s %6s %9.2f us" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
