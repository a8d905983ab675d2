cd $GRAFT_REPO_ROOT
{
for v in g1btrace; do echo "=== $v"; WGFLOW_LIB=variants/lib_$v.so python tools/experiments/g192_trace.py 2>&1 | grep -v "Warn\|WeightNorm\|amdgpu.ids"; done
bash tools/experiments/run_variants.sh g0 g1b g1bnoload g0 g1b
echo "=== parity (g1b)"; WGFLOW_LIB=variants/lib_g1b.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_batch or c2_single" 2>&1 | tail -3
} > gpurun_out/r05d.txt 2>&1
cat gpurun_out/r05d.txt
