cd $GRAFT_REPO_ROOT
for i in 1 2; do
python -m pytest tests/test_gpu_parity.py -x -q -k "layer_launch_of_the_training or repeated_steps or layer_as_one_launch or one_launch_layer or trainer_step" > gpurun_out/r06p_tests.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06p_tests.log | tail -3
done
