cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  WGFLOW_LIB=$R/variants/lib_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pk_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-inverse --no-extra > /dev/null 2>&1
  f=$(ls -t $(find $R/gpurun_out/prof_pk_$v -name "*kernel_stats.csv") | head -1)
  printf "%s: " $v; grep -a "packimg\|rownorm" $f | awk -F, '{printf "%s %.1f us x %s  ", substr($1,2,14), $4/1000, $2}'; echo
done
