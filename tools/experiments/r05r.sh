# round 5: WaveFlow's small kernels with their loads batched (wf_upsample_bwd_kernel, wf_couple_kernel): parity, then the step
cd $GRAFT_REPO_ROOT
{
echo "=== parity (waveflow)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -k "wf or waveflow or wn2d" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
for rep in 1 2; do python bench.py --model waveflow --steps 10 --warmup 3 --no-box --no-inverse 2>/dev/null | python tools/experiments/bench_rows.py | head -3; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05r -- python3 $GRAFT_REPO_ROOT/bench.py --model waveflow --no-inverse --no-box --steps 3 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_r05r/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-60s %6s %9.2f us" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
} > gpurun_out/r05r.txt 2>&1
cat gpurun_out/r05r.txt
