# usage: bash tools/experiments/prof_any.sh <tag> <script.py> [args]: rocprofv3 kernel-trace stats of any python script -> top kernels
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/$@ > $R/gpurun_out/prof_$tag.log 2>&1
f=$(ls -t $(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv") | head -1)
head -22 $f | cut -d, -f1-5 | cut -c1-150
