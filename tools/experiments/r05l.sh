# round 5: the one-launch layer on 256 x 192 tiles (convlayer16g_kernel): parity, timeline, the training step, A/B by env
cd $GRAFT_REPO_ROOT
{
echo "=== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "flattened or wide_batch or c2_single or repeated_steps" 2>&1 | tail -3
echo "=== trace"; WGFLOW_LIB=variants/lib_gltrace.so python tools/experiments/g192_layer_trace.py 2>&1 | grep product
for rep in 1 2 3; do for g in 0 1; do printf "WG_LAYER_G=%s " $g; WG_LAYER_G=$g python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --no-inverse --no-box 2>/dev/null | python tools/experiments/bench_rows.py | head -2; done; done
} > gpurun_out/r05l.txt 2>&1
cat gpurun_out/r05l.txt
