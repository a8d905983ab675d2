"""Phase timeline and in-kernel clock of wgrad16t_kernel (developer experiment): needs a -DWG_DBG_TRACE build of libwgflow.so.

    WGFLOW_LIB=.../variants/lib_trace.so python tools/experiments/wgrad_trace.py [out.json]

Runs coupling forward + backward at the C2 shape and reads the stamps of the LAST weight-gradient launch: per workgroup (= CU) start,
first barrier, end of the main loop and of the slab store of each of its items; shader cycles / wall time = the clock held; the spread of
the main-loop end inside the sets of workgroups that share operands (slots 0..13, 14..27, 28..31 of an XCD in phase A)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import constant_memory_waveglow_amd as cm          # noqa: E402
from constant_memory_waveglow_amd import _lib      # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = cm.AffineCouplingBlock(cm.WN, False, in_channels=4, aux_channels=80, zero_init=False, dilation_channels=256,
                             residual_channels=256, skip_channels=256, depth=8).to(dev)
x = torch.rand(24, 8, 2000, device=dev) * 2 - 1
y = torch.randn(24, 80, 2000, device=dev)
for _ in range(4):
    xx = x.clone().requires_grad_(True)
    z, ls = blk(xx, y)
    (z.sum() + ls.sum()).backward()
torch.cuda.synchronize()
L = _lib.lib()
n = 512 * 16
wb, cb = (C.c_ulonglong * n)(), (C.c_ulonglong * n)()
L.wg_dbg_trace_read.argtypes = [C.c_void_p, C.c_int]
L.wg_dbg_trace_read_cycles.argtypes = [C.c_void_p, C.c_int]
assert L.wg_dbg_trace_read(wb, n) == 0 and L.wg_dbg_trace_read_cycles(cb, n) == 0
wall = np.frombuffer(wb, dtype=np.uint64).reshape(512, 16).astype(np.float64)[:256, 8:14]      # 10 ns ticks
cyc = np.frombuffer(cb, dtype=np.uint64).reshape(512, 16).astype(np.float64)[:256, 8:14]
t = (wall - wall[:, 0].min()) / 100.0                                                           # us
names = ["start", "bar0", "loopA", "slabA", "loopB", "slabB"]
print("workgroup (xcd, slot)  " + " ".join("%8s" % s for s in names))
for w in (0, 8, 104, 112, 216, 224, 248, 1, 255):
    print("%4d (%d, %2d)          " % (w, w & 7, w >> 3) + " ".join("%8.1f" % v for v in t[w]))
d = np.diff(t, axis=1)
print("phase durations, mean / std over the 256 workgroups (us): " + " ".join("%s %.1f/%.1f" % (s, m, sd) for s, m, sd in zip(names[1:], d.mean(0), d.std(0))))
ghz = np.diff(cyc, axis=1) / np.maximum(np.diff(wall, axis=1), 1) / 10.0
print("clock held per phase (GHz, median): " + " ".join("%s %.2f" % (s, v) for s, v in zip(names[1:], np.median(ghz, axis=0))))
whole = float(np.median((cyc[:, 5] - cyc[:, 0]) / (wall[:, 5] - wall[:, 0]) / 10.0))
print("whole kernel: %.2f GHz; launch %.1f us" % (whole, t[:, 5].max()))
slot, xcd = np.arange(256) >> 3, np.arange(256) & 7
spread = []
for x_ in range(8):
    for lo, hi in ((0, 14), (14, 28), (28, 32)):
        m = (xcd == x_) & (slot >= lo) & (slot < hi)
        spread.append(t[m, 2].max() - t[m, 2].min())
print("end of phase A inside a set: max - min, mean over the 24 sets %.1f us, worst %.1f us" % (np.mean(spread), np.max(spread)))
cycA = float(np.median(cyc[:, 2] - cyc[:, 1]))
print("phase A main loop: %.0f cycles (median)" % cycA)
if len(sys.argv) > 1:
    json.dump({"what": "wgrad16t_kernel, last launch of four coupling forward + backward passes at the C2 shape (random data): phase stamps per "
                       "workgroup, shader cycles (s_memtime) / wall time (s_memrealtime) = the clock held; -DWG_DBG_TRACE build",
               "lib": _lib.LIB_PATH, "launch_us": round(float(t[:, 5].max()), 1), "whole_kernel_ghz": round(whole, 3),
               "phase_us_mean": {s: round(float(v), 1) for s, v in zip(names[1:], d.mean(0))},
               "phase_us_std": {s: round(float(v), 1) for s, v in zip(names[1:], d.std(0))},
               "phase_ghz": {s: round(float(v), 3) for s, v in zip(names[1:], np.median(ghz, axis=0))},
               "phaseA_cycles_median": cycA, "set_end_spread_us_mean": round(float(np.mean(spread)), 2),
               "set_end_spread_us_max": round(float(np.max(spread)), 2)}, open(sys.argv[1], "w"), indent=1)
