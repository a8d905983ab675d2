"""Phase timeline of the gate conv in single-utterance synthesis (developer experiment): needs a
-DWG_DBG_TRACE,WG_DBG_TRACE_SMALL build (stamps convgemm16h, or with WG_OPT_NO_HTILE the 128 x 64 form of convgemm16q).

    WGFLOW_LIB=.../variants/lib_trsmall.so python tools/experiments/infer_trace.py [frames]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
from constant_memory_waveglow_amd import _lib      # noqa: E402

dev = torch.device("cuda:0")
m = bench.build_model(dev)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 63
h = torch.randn(1, 80, frames, device=dev)
with torch.no_grad():
    for _ in range(5):
        m.infer(h, 0.6)
torch.cuda.synchronize()
L = _lib.lib()
N = 512
buf = (C.c_ulonglong * (N * 16))()
cb = (C.c_ulonglong * (N * 16))()
L.wg_dbg_trace_read.argtypes = [C.c_void_p, C.c_int]
L.wg_dbg_trace_read_cycles.argtypes = [C.c_void_p, C.c_int]
assert L.wg_dbg_trace_read(buf, N * 16) == 0 and L.wg_dbg_trace_read_cycles(cb, N * 16) == 0
wall = np.frombuffer(buf, dtype=np.uint64).reshape(N, 16).astype(np.float64)      # 10 ns ticks
cyc = np.frombuffer(cb, dtype=np.uint64).reshape(N, 16).astype(np.float64)
nwg = int((wall[:, 0] > 0).sum())
order = [8, 0, 1, 2, 3]
names = ["entry", "init", "bar0", "mainloop", "epilogue"]

w = wall[:nwg][:, order] / 100.0
w -= w[:, 0].min()
c = cyc[:nwg][:, order]
print("%d workgroups; us since the first workgroup's entry" % nwg)
print("wg        " + " ".join("%9s" % n for n in names))
for i in [0, 1, 2, 3, 8, 31, 32, 64, 127, 128, 255]:
    if i < nwg:
        print("%5d     " % i + " ".join("%9.2f" % v for v in w[i]))
d = np.diff(w, axis=1)
dc = np.diff(c, axis=1)
print("phase us (mean)  : " + " ".join("%s %.2f" % (n, v) for n, v in zip(names[1:], d.mean(axis=0))))
print("phase cycles (med): " + " ".join("%s %.0f" % (n, v) for n, v in zip(names[1:], np.median(dc, axis=0))))
print("clock GHz (med)  : " + " ".join("%s %.2f" % (n, v) for n, v in zip(names[1:], np.median(dc / np.maximum(d * 1000.0, 1), axis=0))))
print("entry spread %.2f us, last epilogue end %.2f us" % (w[:, 0].max(), w[:, -1].max()))
