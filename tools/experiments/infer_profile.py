"""Single-utterance synthesis (h[1, 80, 63] -> 16 128 samples) repeated: for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
m = bench.build_model(dev)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 63
h = torch.randn(1, 80, frames, device=dev)
with torch.no_grad():
    for _ in range(3):
        m.infer(h, 0.6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        x = m.infer(h, 0.6)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print("%d samples: %.3f ms per call = %.2f MHz" % (x.numel(), dt * 1e3, x.numel() / dt / 1e6))
