import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["wf_bench.py", "--steps", "3", "--warmup", "1"]
import torch, time, json
import constant_memory_waveglow_amd as cm
CFG = dict(flows=8, n_group=64, n_mels=80, dilation_channels=64, residual_channels=64, skip_channels=64)
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **CFG)
with torch.no_grad():
    for wn in m.WNs: wn.end.weight.normal_(0.0, 0.02)
m = m.to(dev); crit = cm.WaveGlowLoss(1.0)
x = torch.rand(12, 16000, device=dev) * 2 - 1; h = torch.randn(12, 80, 63, device=dev)
for _ in range(4):
    m.zero_grad(set_to_none=True); z, ld = m(x, h); crit(z, ld).backward()
torch.cuda.synchronize()
