"""Single-call synthesis latency (sync, call, sync -- what inference.py:50-56 times) vs back-to-back throughput, and where the host
side of a call goes.   python tools/experiments/infer_latency.py [frames]   (WG_GRAPHS=1: hipGraph replay of wg_inverse)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
m = bench.build_model(dev)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 63
h = torch.randn(1, 80, frames, device=dev)
lat, enq = [], []
with torch.no_grad():
    for _ in range(3):
        m.infer(h, 0.6)
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = m.infer(h, 0.6)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        lat.append(t2 - t0); enq.append(t1 - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        x = m.infer(h, 0.6)
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / 20
lat.sort(); enq.sort()
print("graphs=%s  %d samples: single call %.3f ms (median; min %.3f), of which the call itself (host enqueue) %.3f ms; back to back %.3f ms per call"
      % (os.environ.get("WG_GRAPHS", "0"), x.numel(), lat[15] * 1e3, lat[0] * 1e3, enq[15] * 1e3, thr * 1e3))
# host-side pieces
import cProfile, pstats
with torch.no_grad():
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(10):
        m.infer(h, 0.6)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(14)
