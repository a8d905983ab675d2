# round 5: phase stamps + timing variants of convgemm16g_kernel
cd $GRAFT_REPO_ROOT
{
for v in gtrace gtrace4; do echo "=== $v"; WGFLOW_LIB=variants/lib_$v.so python tools/experiments/g192_trace.py 2>&1 | grep -v "Warn\|WeightNorm\|amdgpu.ids"; done
echo "=== old (WG_G192=0)"; WG_G192=0 python tools/kbench.py --iters 4 --fwd-only 2>&1 | grep -v "Warn\|WeightNorm\|amdgpu.ids" | tail -4
bash tools/experiments/run_variants.sh g0 gnoload gnobar g0
echo "=== parity (default lib)"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_batch or c2_single" 2>&1 | tail -3
} > gpurun_out/r05c.txt 2>&1
cat gpurun_out/r05c.txt
