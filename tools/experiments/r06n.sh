# tanh from the gate's S-plane (WG_TW_FROM_GATE): parity subset incl. the float64 comparison at the headline shape, then a same-box A/B
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -s -k "wide_batch or shape_sweep or flattened or c2_full or full_size_properties or trainer_step or stored_activation or repeated_steps or model_step_vs_oracle or waveflow_model or float64_oracle or wsrglow_model" > gpurun_out/r06n_tests.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06n_tests.log | tail -3; grep -E "headline shape vs float64|WaveFlow 12 x 16000" gpurun_out/r06n_tests.log
for rep in 1 2; do for sw in 0 1; do
  printf "WG_TW_FROM_GATE=%s " $sw
  WG_TW_FROM_GATE=$sw python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --no-inverse --no-box 2>/dev/null | python tools/experiments/bench_rows.py | head -5
done; done 2>&1 | tee gpurun_out/r06n_ab.txt
