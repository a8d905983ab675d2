# round 5: the 64 x 64-tile LDS-DMA kernel (convgemm16m_kernel) in synthesis: parity, then latency with and without it (env switch)
cd $GRAFT_REPO_ROOT
{
echo "=== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "inverse or infer or c2_single or wf_ or waveflow_model or half_inference or layer_alone or wn2d or wn_forward" 2>&1 | tail -5
for m in 0 1 0 1; do echo "WG_M16=$m"; WG_M16=$m python tools/experiments/infer_profile.py 63 2>&1 | tail -1; WG_M16=$m python tools/experiments/infer_profile.py 862 2>&1 | tail -1; done
for m in 0 1; do echo "WG_M16=$m waveflow"; WG_M16=$m python tools/experiments/wf_infer_profile.py 2>&1 | tail -3; done
} > gpurun_out/r05o.txt 2>&1
cat gpurun_out/r05o.txt
