R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "noncausal_layer" > gpurun_out/r04ab_pytest.log 2>&1; echo "pytest rc $?"; grep -v amdgpu.ids gpurun_out/r04ab_pytest.log | grep -E "passed|failed|Error|assert" | tail -12
