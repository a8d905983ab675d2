"""WaveFlow inverse at several utterance lengths (the row-by-row launches are latency bound, so kHz grows with the length)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import constant_memory_waveglow_amd as cm
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = cm.WaveFlow(flows=8, n_group=64, n_mels=80, use_conv1x1=False, memory_efficient=False, dilation_channels=64, residual_channels=64, skip_channels=64, bias=False)
with torch.no_grad():
    for wn in m.WNs: wn.end.weight.normal_(0.0, 0.02)
m = m.to(dev)
for frames in (63, 250, 862):
    N = (frames - 1) * 256
    h = torch.randn(1, 80, frames, device=dev); z = torch.randn(1, N, device=dev) * 0.6
    with torch.no_grad():
        m.reverse(z, h); torch.cuda.synchronize(); t = time.perf_counter(); x, _ = m.reverse(z, h); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("samples %7d  %.1f ms  %.1f kHz" % (N, dt * 1e3, N / dt / 1e3), flush=True)
