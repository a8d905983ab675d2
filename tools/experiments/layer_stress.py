"""Stress of the one-launch layer's hand-off (wg_layer16h.h): forward + inverse of a model, fused, many times; every repetition is compared
with the two-launch result (WG_LAYER_FUSION=0) and with the first fused repetition.   python layer_stress.py [micro|c1|c2] [reps]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fill
import constant_memory_waveglow_amd as cm
from constant_memory_waveglow_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "micro"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
cfg = fill.CONFIGS[name]
specs = fill.model_param_specs(cfg)
P = fill.fill_params(specs, name + "/")
kw = dict(cfg)
m = cm.WaveGlow(memory_efficient=True, bias=kw.pop("bias", False), **kw)
m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
m = m.to(dev)
_, N, F = fill.SHAPES[name]
audio, h = fill.inputs(name, 1, N, F, cfg["n_mels"])
x, ht = torch.from_numpy(audio).to(dev), torch.from_numpy(h).to(dev)
def run():
    with torch.no_grad():
        z, ld = m(x.clone(), ht)
        xr, _ = m.reverse(z, ht)
    return z, xr
cold = len(sys.argv) > 3 and sys.argv[3] == "cold"       # the FUSED path runs first in this process (cold caches, code objects, TLBs)
fused_runs = []
def fused_pass():
    os.environ["WG_LAYER_FUSION"] = "1"
    for r in range(reps):
        z, xr = run()
        torch.cuda.synchronize()
        fused_runs.append((z, xr))
t0 = time.time()
before = _lib.lib().wg_stat_layer_launches()
if cold:
    fused_pass()
os.environ["WG_LAYER_FUSION"] = "0"
z0, x0 = run()
z0b, x0b = run()
rep2 = torch.equal(z0, z0b) and torch.equal(x0, x0b)
if not cold:
    t0 = time.time()
    fused_pass()
dt = (time.time() - t0) / reps * 1e3
bad_ref, bad_rep, worst, nan = 0, 0, 0.0, 0
for z, xr in fused_runs:
    e = max(float((z - z0).abs().max()), float((xr - x0).abs().max()))
    if not np.isfinite(e):
        nan += 1
    elif e > 1e-5:
        bad_ref += 1
    worst = max(worst, e if np.isfinite(e) else 0.0)
    if not (torch.equal(z, fused_runs[0][0]) and torch.equal(xr, fused_runs[0][1])):
        bad_rep += 1
print("%s %s (lib %s): two-launch repeatable %s; %d fused repetitions, %d fused launches, %.1f ms per repetition: %d differ from the two-launch result by "
      "> 1e-5 (worst %.2e), %d non-finite, %d differ from the first repetition" % (name, "cold" if cold else "warm", os.path.basename(_lib.LIB_PATH), rep2, reps,
      _lib.lib().wg_stat_layer_launches() - before, dt, bad_ref, worst, nan, bad_rep))
