"""WaveFlow (configs/waveflow_LJ_speech.json: 8 flows, n_group 64, 80 mels, 64 channels, batch 12, segment 16000) forward + NLL +
backward on one MI355X, and the row-by-row inverse: samples/s, ms/step, inverse kHz (developer tool; the headline is bench.py).

    python tools/wf_bench.py [--batch 12] [--segment 16000] [--steps 5] [--warmup 2]
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

CFG = dict(flows=8, n_group=64, n_mels=80, dilation_channels=64, residual_channels=64, skip_channels=64)
# per (row, time) position and layer: 3x3 conv 2*C*2Cd*9, conditioning 2*n_mels*2Cd, W_o 2*Cd*(C+Cs)
FLOP_PER_POS_LAYER = 2 * 64 * 128 * 9 + 2 * 80 * 128 + 2 * 64 * 128


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--segment", type=int, default=16000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **CFG)
    with torch.no_grad():
        for wn in m.WNs:
            wn.end.weight.normal_(0.0, 0.02)
    m = m.to(dev)
    crit = cm.WaveGlowLoss(1.0)
    N = a.segment - a.segment % 64
    frames = N // 256 + 1
    x = torch.rand(a.batch, N, device=dev) * 2 - 1
    h = torch.randn(a.batch, 80, frames, device=dev)

    def step():
        m.zero_grad(set_to_none=True)
        m._engine.packed.key = None                # as in training, where the weights change: re-pack every step
        z, logdet = m(x, h)
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
    pos = a.batch * 63 * (N // 64)
    flop = 3.0 * 8 * 8 * FLOP_PER_POS_LAYER * pos
    out = {"workload": "WaveFlow 64ch, 8 flows, n_group 64, batch %d, segment %d" % (a.batch, N), "ms_per_step": dt * 1e3,
           "samples_per_s": a.batch * N / dt, "algorithmic_tflops": flop / dt / 1e12,
           "params": sum(p.numel() for p in m.parameters()), "loss": float(loss), "mem_gib": torch.cuda.max_memory_allocated() / 2 ** 30}
    with torch.no_grad():
        hc = h[:1]
        zc = torch.randn(1, N, device=dev) * 0.6
        m.reverse(zc, hc)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        xs, _ = m.reverse(zc, hc)
        torch.cuda.synchronize()
        out["inverse_khz_%d" % N] = N / (time.perf_counter() - t1) / 1000.0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
