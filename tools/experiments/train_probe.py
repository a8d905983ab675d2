"""Sanity of the whole training loop (developer experiment): 60 steps of FlowTrainer + FlatAdam on one fixed batch of the
headline network at the reference's learning rate must drive the loss down (0.018 -> -0.92 on an MI355X).  At lr 1e-3 the flow
diverges within three steps -- and so does the upstream reference on CPU with torch.optim.Adam (0.018, -0.119, 5.2e5, ...), so that is
the model, not the engine."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from constant_memory_waveglow_amd.parallel import FlowTrainer, FlatAdam
dev = torch.device("cuda:0")
m = bench.build_model(dev)
tr = FlowTrainer(m, bench.SIGMA)
opt = FlatAdam(tr, lr=1e-4)          # the reference's learning rate (configs/waveglow_LJ_speech.json); 1e-3 diverges within 3 steps
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(8, bench.SEG, device=dev, generator=g) * 2 - 1) * 0.3
h = torch.randn(8, 80, bench.FRAMES, device=dev, generator=g)
ls = []
for i in range(60):
    loss, z, ld = tr.step(x, h)
    ls.append(float(loss))
print("loss first 5:", [round(v, 4) for v in ls[:5]])
print("loss last 5 :", [round(v, 4) for v in ls[-5:]])
assert all(v == v for v in ls) and ls[-1] < ls[0] - 0.3, ls
print("ok: loss decreases monotonically-ish:", ls[0], "->", ls[-1])
