cd $GRAFT_REPO_ROOT
for i in 1 2; do
python -m pytest tests/test_gpu_parity.py -q -x -k "test_weight_gradient_kernel_plans_vs_oracle and 1500" 2>&1 | grep -E "passed|failed|FAILED|AssertionError: |assert 0\." | head -8
done
echo "--- WG_LOWRANK=0"
WG_LOWRANK=0 python -m pytest tests/test_gpu_parity.py -q -x -k "test_weight_gradient_kernel_plans_vs_oracle and 1500" 2>&1 | grep -E "passed|failed|FAILED|AssertionError: |assert 0\." | head -8
echo "--- WG_TW_FROM_GATE=0"
WG_TW_FROM_GATE=0 python -m pytest tests/test_gpu_parity.py -q -x -k "test_weight_gradient_kernel_plans_vs_oracle and 1500" 2>&1 | grep -E "passed|failed|FAILED|AssertionError: |assert 0\." | head -8
echo "--- WG_G192_SPLITK=0"
WG_G192_SPLITK=0 python -m pytest tests/test_gpu_parity.py -q -x -k "test_weight_gradient_kernel_plans_vs_oracle and 1500" 2>&1 | grep -E "passed|failed|FAILED|AssertionError: |assert 0\." | head -8
echo "--- new bias test"
python -m pytest tests/test_gpu_parity.py -q -k "with_biases" 2>&1 | grep -E "passed|failed|FAILED|Error" | head
