"""A/B in one process: wg_forward + loss + wg_backward (separate calls) against wg_train_step (one call, last flow not recomputed)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from constant_memory_waveglow_amd import engine as E
from constant_memory_waveglow_amd.parallel import FlowTrainer
dev = torch.device("cuda:0")
tr = FlowTrainer(bench.build_model(dev), bench.SIGMA)
x = torch.rand(24, bench.SEG, device=dev) * 2 - 1
h = torch.randn(24, 80, bench.FRAMES, device=dev)
eng = tr.model._engine
table = [t.detach() for t in tr.table]
need = [True] * len(table)

def separate():
    z, logdet = eng.run(table, x, h, False)
    loss = E.nll_loss(z, logdet, tr.sigma, True)
    dz, dld = E.nll_loss_backward(z, tr.sigma, True, torch.ones((), device=dev))
    eng.backward(table, z, h, dz, dld, need, False, False, grads_out=tr.grad_views)
    return loss

def fused():
    return tr.step(x, h)[0]

for name, fn in (("separate", separate), ("fused", fused), ("separate", separate), ("fused", fused)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): l = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print("%-9s %.2f ms/step  loss %.6f" % (name, dt * 1e3, float(l)), flush=True)
