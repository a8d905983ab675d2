cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "float64_oracle or sixteen_rows or c2_full or layer_launch_of_the_training or wsrglow_gate_conv_cut or wsrglow_full or wsrglow_timed" > gpurun_out/r06k_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r06k_tests.log | head; grep -E "vs float64|headline shape" gpurun_out/r06k_tests.log
