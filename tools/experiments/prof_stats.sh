# kernel-trace stats of the training step; writes gpurun_out/prof_<tag>/..._kernel_stats.csv
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-inverse --no-extra > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | cut -c1-300
find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -2
