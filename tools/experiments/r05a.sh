# round 5, first run of the 256 x 192-tile LDS-DMA conv kernel (wg_gemm16g.h): LDS-DMA above 64 KB, same-box A/B per kernel class, parity
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
hipcc --offload-arch=gfx950 -O3 -o /tmp/glds_probe tools/experiments/glds_probe.hip && /tmp/glds_probe
echo "=== kbench WG_G192=0"; WG_G192=0 timeout 300 python tools/kbench.py --iters 4
echo "=== kbench WG_G192=1"; WG_G192=1 timeout 300 python tools/kbench.py --iters 4
echo "=== kbench WG_G192=0"; WG_G192=0 timeout 300 python tools/kbench.py --iters 4
echo "=== kbench WG_G192=1"; WG_G192=1 timeout 300 python tools/kbench.py --iters 4
echo "=== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_batch or c2_single or coupling_block_on_shared" 2>&1 | tail -15
} > gpurun_out/r05a.txt 2>&1
tail -60 gpurun_out/r05a.txt
