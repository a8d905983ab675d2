cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "wf8b or wf64b" > gpurun_out/r04ae.txt 2>&1
tail -30 gpurun_out/r04ae.txt
