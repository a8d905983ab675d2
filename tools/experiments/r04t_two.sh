R=$GRAFT_REPO_ROOT
cd $R
python tools/experiments/two_stream_probe.py 8 2>&1 | grep -v "amdgpu.ids\|Warn\|WeightNorm" | tail -3
