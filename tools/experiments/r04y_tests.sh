R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chip_filling or (weight_gradient_kernel_plans and 6-1024)" --durations=5 > gpurun_out/r04y_pytest.log 2>&1; echo "pytest rc $?"; grep -v amdgpu.ids gpurun_out/r04y_pytest.log | tail -12
