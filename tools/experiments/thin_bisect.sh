# usage: bash tools/experiments/thin_bisect.sh v1 v2 ...: average duration of the thin kernels (wg_thin.h) per variant library, from a kernel trace of 2 steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  WGFLOW_LIB=$R/variants/lib_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-inverse --no-extra > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_$v -name "*kernel_stats.csv")
  printf "%s: " $v; grep -a "thin" $f | awk -F, '{printf "%s %.1f us  ", substr($1,7,16), $4/1000}'; echo
done
