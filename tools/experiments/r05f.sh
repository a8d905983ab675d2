cd $GRAFT_REPO_ROOT
{
for v in $TRACES; do echo "=== $v"; WGFLOW_LIB=variants/lib_$v.so python tools/experiments/g192_trace.py 2>&1 | grep -v "Warn\|WeightNorm\|amdgpu.ids"; done
for v in "$@"; do echo "=== $v"; WGFLOW_LIB=variants/lib_$v.so python tools/kbench.py --iters 8 --fwd-only 2>&1 | grep -v "Warn\|WeightNorm\|amdgpu.ids" | tail -4; done
echo "=== parity"; WGFLOW_LIB=variants/lib_$1.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_batch or c2_single" 2>&1 | tail -3
} > gpurun_out/r05f.txt 2>&1
cat gpurun_out/r05f.txt
