"""Does WSRGlow's training step take wgrad16t_kernel (planned launches)?  Prints the launch counter around one step and the step time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import constant_memory_waveglow_amd as cm
from constant_memory_waveglow_amd import _lib
from constant_memory_waveglow_amd.parallel import FlowTrainer

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False).to(dev)
for blk in m.WNs:
    torch.nn.init.normal_(blk.F.end.weight, std=0.01)
tr = FlowTrainer(m, 1.0)
x = torch.rand(12, 8192, device=dev) * 2 - 1
c = torch.rand(12, 4096, device=dev) * 2 - 1
L = _lib.lib()
for _ in range(2):
    tr.step(x, c.clone())
torch.cuda.synchronize()
n0 = L.wg_stat_wgrad16t_launches()
t0 = time.perf_counter()
for _ in range(5):
    tr.step(x, c.clone())
torch.cuda.synchronize()
print("wgrad16t launches per step:", (L.wg_stat_wgrad16t_launches() - n0) / 5, " ms/step: %.2f" % ((time.perf_counter() - t0) / 5 * 1e3))
