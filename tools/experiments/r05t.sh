# end_affine_kernel<8, false> after it gained the SEAM template parameter: per-launch time in training and in synthesis (default env)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05t_a -- python3 $R/tools/experiments/infer_profile.py 63 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05t_b -- python3 $R/bench.py --no-inverse --no-box --steps 3 --warmup 1 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv,glob
for tag in ("a","b"):
    f=sorted(glob.glob('gpurun_out/prof_r05t_%s/**/*kernel_stats.csv'%tag,recursive=True))[-1]
    for r in list(csv.DictReader(open(f))):
        if any(k in r['Name'] for k in ('end_affine','mix_kernel<8','start_fwd','convlayer16g','wgrad16t')): print(tag, "%-60s %6s %9.2f us" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
python bench.py --steps 10 --warmup 3 --no-inverse 2>/dev/null | python tools/experiments/bench_rows.py | head -2
