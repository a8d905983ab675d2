# round 5: the residual product of convlayer16g_kernel with the activation pieces of its first two chunks requested in front of the drain wait (-DWGG_OPT_PRE_B)
cd $GRAFT_REPO_ROOT
{
echo "=== parity with the variant"; WGFLOW_LIB=variants/lib_preb.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "flattened or layer_as_one_launch or c2_full_batch" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
ROWS=2 bash tools/experiments/ab_bench.sh base preb
bash tools/experiments/ab_bench.sh base preb
} > gpurun_out/r05u.txt 2>&1
cat gpurun_out/r05u.txt
