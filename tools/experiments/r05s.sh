# round 5: the inverse direction's seam (affine + 1x1 + next WN.start in one launch): equality with the three launches, parity, time per call
cd $GRAFT_REPO_ROOT
{
python - <<'PY'
import os, torch, bench
dev = torch.device("cuda:0")
m = bench.build_model(dev)
torch.manual_seed(1)
for frames, B in ((63, 1), (63, 3), (200, 2)):
    h = torch.randn(B, 80, frames, device=dev)
    z = torch.randn(B, frames * 256, device=dev) * 0.6
    outs = []
    for sw in ("1", "0"):
        os.environ["WG_INV_SEAM"] = sw
        with torch.no_grad():
            x, ld = m.reverse(z.clone(), h)
        outs.append((x.clone(), ld.clone()))
    print("frames", frames, "B", B, "equal:", torch.equal(outs[0][0], outs[1][0]), torch.equal(outs[0][1], outs[1][1]), float((outs[0][0]-outs[1][0]).abs().max()))
os.environ.pop("WG_INV_SEAM")
PY
echo "=== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or inverse or reverse or infer or full_size or one_launch" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2; do for sw in 1 0; do printf "WG_INV_SEAM=%s " $sw; WG_INV_SEAM=$sw python tools/experiments/infer_profile.py 63 2>/dev/null | tail -1; done; done
} > gpurun_out/r05s.txt 2>&1
cat gpurun_out/r05s.txt
