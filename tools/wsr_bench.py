"""WSRGlow 2x (configs/wsrglow_vctk_2x.json: 12 flows, n_group 16, 3659 conditioning channels, WN 256ch x 8 layers, batch 12,
segment 8192) forward + NLL + constant-memory backward on one MI355X: samples/s and ms/step (developer tool; the headline
benchmark is bench.py).

    python tools/wsr_bench.py [--batch 12] [--segment 8192] [--steps 5] [--warmup 2]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import constant_memory_waveglow_amd as cm          # noqa: E402

FLOP_PER_FLOW_STEP = 2 * (3659 * 4096 + 8 * 3 * 256 * 512 + 7 * 256 * 512 + 256 * 256)   # V + W + W_o per time step (start/end small)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--segment", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False)
    with torch.no_grad():
        for blk in m.WNs:
            blk.F.end.weight.normal_(0.0, 0.02)
    m = m.to(dev)
    crit = cm.WaveGlowLoss(1.0)
    x = torch.rand(a.batch, a.segment, device=dev) * 2 - 1
    c = (torch.rand(a.batch, a.segment // 2, device=dev) * 2 - 1) * 0.9

    def step():
        m.zero_grad(set_to_none=True)
        m._engine.packed.key = None                # as in training, where the weights change: re-pack every step
        z, logdet = m(x, c.clone())
        loss = crit(z, logdet)
        loss.backward()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    T = a.segment // 16
    flop = 3.0 * 12 * FLOP_PER_FLOW_STEP * T * a.batch
    print(json.dumps({"workload": "WSRGlow 2x, batch %d, segment %d" % (a.batch, a.segment), "ms_per_step": dt * 1e3,
                      "samples_per_s": a.batch * a.segment / dt, "algorithmic_tflops": flop / dt / 1e12,
                      "params": sum(p.numel() for p in m.parameters()), "loss": float(loss),
                      "mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
