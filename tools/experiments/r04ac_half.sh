R=$GRAFT_REPO_ROOT
cd $R
for c in 0 128 192; do
  if [ $c = 0 ]; then unset WG_CUS; else export WG_CUS=$c; fi
  echo "WG_CUS=$c"; python tools/experiments/two_stream_probe.py 6 2>&1 | grep -v "amdgpu.ids\|Warn\|WeightNorm" | tail -2
done
