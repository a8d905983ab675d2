"""Per-kernel-class timing of ONE flow (coupling block at the C2 shape) through the C ABI, for A/B builds.

    WGFLOW_LIB=/path/to/variant.so python tools/kbench.py [--iters 5] [--B 24] [--T 2000]

Prints avg ms per launch and TFLOP/s (hardware FLOPs incl. padding are NOT counted: algorithmic FLOPs) per kernel class.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import constant_memory_waveglow_amd as cm          # noqa: E402
from constant_memory_waveglow_amd import _lib      # noqa: E402

NAMES = ["conv_store", "conv_gate", "conv_resskip", "conv_dgate", "wgrad", "layer"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--B", type=int, default=24)
    ap.add_argument("--T", type=int, default=2000)
    ap.add_argument("--ch", type=int, default=256)
    ap.add_argument("--fwd-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    blk = cm.AffineCouplingBlock(cm.WN, False, in_channels=4, aux_channels=80, zero_init=False, dilation_channels=a.ch,
                                 residual_channels=a.ch, skip_channels=a.ch, depth=8).to(dev)
    x = torch.rand(a.B, 8, a.T, device=dev) * 2 - 1
    y = torch.randn(a.B, 80, a.T, device=dev)
    L = _lib.lib()
    C_, Bt = a.ch, a.B * a.T
    flops = {  # algorithmic FLOPs per launch of the BIG instance of each class
        "conv_gate": 2.0 * (3 * C_ + 80) * 2 * C_ * Bt, "conv_resskip": 2.0 * C_ * 2 * C_ * Bt,
        "conv_dgate": 2.0 * 2 * C_ * C_ * Bt, "conv_store": 2.0 * 3 * 2 * C_ * C_ * Bt, "wgrad": 2.0 * 2 * C_ * (3 * C_ + 80) * Bt,
        "layer": 2.0 * (3 * C_ + 80) * 2 * C_ * Bt + 2.0 * C_ * C_ * Bt}

    def run():
        xx = x.clone().requires_grad_(True)
        z, ls = blk(xx, y)
        if not a.fwd_only:
            (z.sum() + ls.sum()).backward()

    run()
    torch.cuda.synchronize()
    print("lib:", _lib.LIB_PATH)
    for kid, name in enumerate(NAMES):
        if a.fwd_only and name in ("conv_dgate", "wgrad"):
            continue
        t = L.wg_timer_create(kid, 4096)
        L.wg_timer_attach(t)
        for _ in range(a.iters):
            run()
        torch.cuda.synchronize()
        L.wg_timer_attach(None)
        n = L.wg_timer_count(t)
        buf = (C.c_float * n)()
        L.wg_timer_read(t, buf, n)
        L.wg_timer_destroy(t)
        ms = np.frombuffer(buf, dtype=np.float32).copy()
        if ms.size == 0:
            print("%-13s no launches" % name)
            continue
        big = ms[ms > 0.5 * ms.max()]
        print("%-13s launches/iter %3d  total %.3f ms/iter  big-instance avg %.4f ms (n=%d) -> %.1f TF" % (
            name, n // a.iters, ms.sum() / a.iters, big.mean(), big.size, flops[name] / (big.mean() * 1e-3) / 1e12))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("whole flow (fwd%s): %.3f ms/iter" % ("" if a.fwd_only else "+bwd", e0.elapsed_time(e1) / a.iters))


if __name__ == "__main__":
    main()
