"""Would the training step gain from running as two concurrent half-batch chains (stream A multiplies while stream B moves bytes)?
Two model replicas (same weights), each with its own engine / workspace / gradient buffer, batch 12 each, stepped on two streams at once,
against one replica at batch 24.   python tools/experiments/two_stream_probe.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from constant_memory_waveglow_amd.parallel import FlowTrainer
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m0 = bench.build_model(dev)
t0 = FlowTrainer(m0, bench.SIGMA)
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.rand(24, bench.SEG, device=dev, generator=g) * 2 - 1
h = torch.randn(24, 80, bench.FRAMES, device=dev, generator=g)
def timeit(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
one = timeit(lambda: t0.step(x, h), steps)
ma, mb = bench.build_model(dev), bench.build_model(dev)
ta, tb = FlowTrainer(ma, bench.SIGMA), FlowTrainer(mb, bench.SIGMA)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
xa, ha, xb, hb = x[:12].contiguous(), h[:12].contiguous(), x[12:].contiguous(), h[12:].contiguous()
def two():
    cur = torch.cuda.current_stream(dev)
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        ta.step(xa, ha)
    with torch.cuda.stream(sb):
        tb.step(xb, hb)
    cur.wait_stream(sa); cur.wait_stream(sb)
    # the full-batch gradient = the mean of the halves' (each normalised by its own 12 N)
    torch.add(ta.fg.flat, tb.fg.flat, out=ta.fg.flat).mul_(0.5)
both = timeit(two, steps)
half = timeit(lambda: ta.step(xa, ha), steps)
print("batch 24 in one chain %.2f ms; two concurrent chains of 12: %.2f ms; one chain of 12 alone: %.2f ms (x2 = %.2f)" % (one, both, half, 2 * half))
# same gradients?
t0.step(x, h); two(); torch.cuda.synchronize()
d = float((t0.fg.flat - ta.fg.flat).abs().max() / t0.fg.flat.abs().max())
print("max |grad difference| / max |grad| = %.2e" % d)
