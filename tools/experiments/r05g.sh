cd $GRAFT_REPO_ROOT
{
for v in gsttrace; do
echo "=== $v timeline"; WGFLOW_LIB=variants/lib_$v.so python tools/experiments/g192_trace.py 2>&1 | grep "timeline\|residual"
echo "=== $v by HIP events"; WGFLOW_LIB=variants/lib_$v.so python tools/kbench.py --iters 8 --fwd-only 2>&1 | grep "conv_gate\|conv_store"
done
} > gpurun_out/r05g.txt 2>&1
cat gpurun_out/r05g.txt
