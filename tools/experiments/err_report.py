"""Error margins of the HIP engine against the reference goldens / the oracle (developer tool; prints, asserts nothing).

    [WGFLOW_LIB=variants/lib_x.so] python tools/experiments/err_report.py

For micro / c1 (oracle + golden) and c2 (golden): max |dz|, |dloss|, worst parameter-gradient error relative to the tensor's max."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fill                                         # noqa: E402
import constant_memory_waveglow_amd as cm          # noqa: E402
from oracle import wg_oracle as orc                # noqa: E402

dev = torch.device("cuda:0")
for name in ("micro", "c1", "c2", "c1x9"):
    wide = name == "c1x9"                       # C1 at batch 9: 4 608 columns, the size from which the one-product skip / S-only chains are used
    if wide:
        name = "c1"
    cfg = fill.CONFIGS[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    B, N, F = fill.SHAPES[name]
    if wide:
        B = 9
    audio, h = fill.inputs(name + ("x9" if wide else ""), B, N, F, cfg["n_mels"])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "model_%s.npz" % name))
    if wide:
        r64 = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True, double=True)
        gold = dict(z=r64["z"], loss=r64["loss"], dh=r64["dh"])
    m = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    x, ht = torch.from_numpy(audio).to(dev), torch.from_numpy(h).to(dev).requires_grad_(True)
    z, ld = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, ld)
    loss.backward()
    ez = float(np.abs(z.detach().cpu().numpy() - gold["z"]).max())
    el = abs(float(loss) - float(gold["loss"]))
    named = dict(m.named_parameters())
    worst, wname = 0.0, ""
    ref = (r64 if wide else orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, double=True)) if name != "c2" else None
    for i, (n, _, _) in enumerate(specs):
        g = named[n].grad.cpu().numpy()
        if g[0].size == 1 and n.endswith("weight_v"):
            continue
        if ref is not None:
            e = float(np.abs(g - ref["grads"][i]).max() / max(np.abs(ref["grads"][i]).max(), 1e-30))
        else:
            nh = min(g.size, gold["grad_head"].shape[1])
            e = float(np.abs(g.ravel()[:nh] - gold["grad_head"][i][:nh]).max() / max(float(gold["grad_max"][i]), 1e-30))
        if e > worst:
            worst, wname = e, n
    edh = float(np.abs(ht.grad.cpu().numpy() - gold["dh"]).max() / np.abs(gold["dh"]).max())
    print("%-6s |dz| %.2e  |dloss| %.2e  dh rel %.2e  worst grad rel %.2e (%s)   [bars: 1e-4, 1e-6, 1e-4, 1e-4]" % (name + ("x9" if wide else ""), ez, el, edh, worst, wname))
