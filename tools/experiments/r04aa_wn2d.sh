R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wn2d_alone" > gpurun_out/r04aa_pytest.log 2>&1; echo "pytest rc $?"; grep -v amdgpu.ids gpurun_out/r04aa_pytest.log | tail -25
