cd $GRAFT_REPO_ROOT
ROWS=6 bash tools/experiments/ab_bench.sh base ni1dg
