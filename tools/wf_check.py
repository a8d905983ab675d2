"""Developer check of the WaveFlow HIP path against the oracle (prints errors instead of asserting)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fill                                          # noqa: E402
from oracle import wf_oracle as wfo                  # noqa: E402
import constant_memory_waveglow_amd as cm            # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for name in sys.argv[1:] or ["wf8", "wf64"]:
        cfg = fill.WF_CONFIGS[name]
        B, N, F = fill.WF_SHAPES[name]
        specs = fill.waveflow_param_specs(cfg)
        P = fill.fill_params(specs, name + "/")
        audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
        ref = wfo.train_step(wfo.make_config(**cfg), fill.table(specs, P), audio, mel, fill.SIGMA, need_dmel=True)
        m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
        m = m.to(dev)
        ht = torch.from_numpy(mel).to(dev).requires_grad_(True)
        z, logdet = m(torch.from_numpy(audio).to(dev), ht)
        print(name, "z", float(np.abs(z.detach().cpu().numpy() - ref["z"]).max()), "logdet", logdet.detach().cpu().numpy(), ref["logdet"], flush=True)
        loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
        loss.backward()
        print("  loss", float(loss), ref["loss"], "dmel", float(np.abs(ht.grad.cpu().numpy() - ref["dmel"]).max() / np.abs(ref["dmel"]).max()), flush=True)
        named = dict(m.named_parameters())
        worst = []
        for i, (n, _, _) in enumerate(specs):
            g = named[n].grad.cpu().numpy()
            e = float(np.abs(g - ref["grads"][i]).max() / max(np.abs(ref["grads"][i]).max(), 1e-30))
            worst.append((e, n))
        worst.sort(reverse=True)
        print("  grads worst:", worst[:6], flush=True)
        with torch.no_grad():
            x, ld = m.reverse(torch.from_numpy(ref["z"]).to(dev), ht.detach())
        print("  inverse", float(np.abs(x.cpu().numpy() - audio).max()), ld.cpu().numpy(), -ref["logdet"], flush=True)


if __name__ == "__main__":
    main()
