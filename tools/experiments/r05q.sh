# round 5: WSRGlow's gate conv cut along K (convgemm16g_kernel<WGG_EPI_PART> + gate_finish16g_kernel): parity, then the step with and without
cd $GRAFT_REPO_ROOT
{
echo "=== parity (wsr)"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -k "wsr" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
for rep in 1 2; do for g in 0 1; do printf "WG_G192_SPLITK=%s " $g; WG_G192_SPLITK=$g python bench.py --model wsrglow --steps 10 --warmup 3 --no-box 2>/dev/null | python tools/experiments/bench_rows.py | head -4; done; done
} > gpurun_out/r05q.txt 2>&1
cat gpurun_out/r05q.txt
