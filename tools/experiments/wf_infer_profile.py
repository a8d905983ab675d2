"""WaveFlow row-by-row synthesis of one utterance, repeated: for rocprofv3 --kernel-trace.   python wf_infer_profile.py [samples] [calls]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import constant_memory_waveglow_amd as cm
sys.path.insert(0, os.path.join(ROOT, "tools"))
from wf_bench import CFG
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **CFG)
with torch.no_grad():
    for wn in m.WNs:
        wn.end.weight.normal_(0.0, 0.02)
m = m.to(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16128
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N -= N % 64
h = torch.randn(1, 80, N // 256 + 1, device=dev)
z = torch.randn(1, N, device=dev) * 0.6
with torch.no_grad():
    m.reverse(z, h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        x, _ = m.reverse(z, h)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / calls
print("WaveFlow inverse %d samples: %.2f ms per call = %.1f kHz" % (N, dt * 1e3, N / dt / 1e3))
