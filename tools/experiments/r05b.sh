# round 5: timing variants of convgemm16g_kernel (forward launches of one coupling block, same box)
cd $GRAFT_REPO_ROOT
{
echo "=== old (WG_G192=0)"; WG_G192=0 python tools/kbench.py --iters 4 --fwd-only 2>&1 | grep -v "Warn\|WeightNorm\|amdgpu.ids" | tail -4
bash tools/experiments/run_variants.sh "$@"
echo "=== parity (default lib)"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_batch or c2_single" 2>&1 | tail -3
} > gpurun_out/r05b.txt 2>&1
cat gpurun_out/r05b.txt
