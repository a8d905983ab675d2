"""Phase timeline of the gate conv (developer experiment): needs a -DWG_DBG_TRACE build of libwgflow.so.

    WGFLOW_LIB=.../variants/trace.so python tools/experiments/conv_trace.py

Runs one coupling forward at the C2 shape, then prints per workgroup slot: start, first barrier, and for each of its tiles the
end of the main loop and the end of the epilogue (us, relative to the earliest workgroup start of the LAST gate-conv launch)."""
import ctypes as C
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
with torch.no_grad():
    for _ in range(3):
        blk(x.clone(), y)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (512 * 16))()
L.wg_dbg_trace_read.argtypes = [C.c_void_p, C.c_int]
assert L.wg_dbg_trace_read(buf, 512 * 16) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.float64) / 100.0     # us
NWG = int((t[:, 0] > 0).sum())                 # workgroups of the launch (512 with two per CU, 256 for the one-per-CU 256x128 form)
t = t[:NWG]
t0 = t[:, 0].min()
t = t - t0
names = ["start", "bar0", "ml0", "ep0", "ml1", "ep1", "ml2", "ep2"]
print("workgroup   " + " ".join("%7s" % n for n in names))
for w in [w for w in list(range(0, 16)) + [100, 255, 256, 400, 511] if w < NWG]:
    print("%9d   " % w + " ".join("%7.1f" % v for v in t[w, :8]))
d = np.diff(t[:, :8], axis=1)
print("mean phase durations (us): " + " ".join("%s %.1f" % (n, v) for n, v in zip(names[1:], d.mean(axis=0))))
print("std                      : " + " ".join("%s %.1f" % (n, v) for n, v in zip(names[1:], d.std(axis=0))))
print("last epilogue end: mean %.1f max %.1f" % (t[:, 7].mean(), t[:, 7].max()))
if hasattr(L, "wg_dbg_trace_read_cycles"):
    cb = (C.c_ulonglong * (512 * 16))()
    L.wg_dbg_trace_read_cycles.argtypes = [C.c_void_p, C.c_int]
    assert L.wg_dbg_trace_read_cycles(cb, 512 * 16) == 0
    cyc = np.frombuffer(cb, dtype=np.uint64).reshape(512, 16).astype(np.float64)[:NWG]
    wall = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.float64)[:NWG]       # 10 ns ticks
    dc, dw = np.diff(cyc[:, :8], axis=1), np.diff(wall[:, :8], axis=1)
    ghz = dc / np.maximum(dw, 1) / 10.0
    print("main loop cycles per tile (median): " + " ".join("%s %.0f" % (n, v) for n, v in zip(names[1:], np.median(dc, axis=0))))
    print("clock held per phase (GHz, median over workgroups): " + " ".join("%s %.2f" % (n, v) for n, v in zip(names[1:], np.median(ghz, axis=0))))
    nwg = NWG
    whole = float(np.median((cyc[:nwg, 7] - cyc[:nwg, 0]) / (wall[:nwg, 7] - wall[:nwg, 0]) / 10.0))
    print("whole kernel: %.2f GHz (%d workgroups)" % (whole, nwg))
    if len(sys.argv) > 1:                                   # python conv_trace.py out.json: the numbers as a small JSON record
        import json
        json.dump({"what": "in-kernel clock of the gate conv (last launch of three coupling forwards at the C2 shape, random data): "
                           "shader cycles (s_memtime) / wall time (s_memrealtime, 100 MHz) per phase, median over workgroups; "
                           "-DWG_DBG_TRACE build of the default kernels, tools/experiments/conv_trace.py",
                   "lib": _lib.LIB_PATH, "workgroups": nwg, "whole_kernel_ghz": round(whole, 3),
                   "phase_ghz": {n: round(float(v), 3) for n, v in zip(names[1:], np.median(ghz[:nwg], axis=0))},
                   "phase_us_mean": {n: round(float(v), 2) for n, v in zip(names[1:], d[:nwg].mean(axis=0))},
                   "launch_us": round(float(t[:nwg, 7].max()), 1)}, open(sys.argv[1], "w"), indent=1)
