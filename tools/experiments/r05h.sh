# round 5: the training step with and without the 256 x 192-tile kernel (same build, env switch), interleaved
cd $GRAFT_REPO_ROOT
{
for rep in 1 2; do for g in 0 1; do printf "WG_G192=%s " $g; WG_G192=$g python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --no-inverse 2>/dev/null | python tools/experiments/bench_rows.py; done; done
} > gpurun_out/r05h.txt 2>&1
cat gpurun_out/r05h.txt
