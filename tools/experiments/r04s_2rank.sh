R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu > gpurun_out/r04s_pytest.log 2>&1; echo "pytest rc $?"; grep -v "amdgpu.ids" gpurun_out/r04s_pytest.log | tail -15
