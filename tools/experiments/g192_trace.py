"""Phase stamps of convgemm16g_kernel<EPI_GATE_SO> (developer experiment): needs a -DWG_DBG_TRACE build of libwgflow.so.

    WGFLOW_LIB=variants/lib_gtrace.so python tools/experiments/g192_trace.py

Runs coupling forwards at the C2 shape, then prints (median over the 256 workgroups of the LAST gate-conv launch, wave WGG_TRACE_WAVE):
the cycles between the stamps inside one chunk, the tile-level timeline and the clock held."""
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
    for _ in range(40):
        blk(x.clone(), y)
torch.cuda.synchronize()
L = _lib.lib()
N = 512 * 16
wb, cb = (C.c_ulonglong * N)(), (C.c_ulonglong * N)()
L.wg_dbg_trace_read.argtypes = [C.c_void_p, C.c_int]
L.wg_dbg_trace_read_cycles.argtypes = [C.c_void_p, C.c_int]
assert L.wg_dbg_trace_read(wb, N) == 0 and L.wg_dbg_trace_read_cycles(cb, N) == 0
wall_all = np.frombuffer(wb, dtype=np.uint64).reshape(512, 16).astype(np.float64) / 100.0       # us
wall = wall_all[:256]
res = wall_all[256:]                                  # the last residual conv (EPI_STORE_SO) launch: the one in front of the last gate conv
print("residual conv in front: entry %.1f .. exit %.1f us before the gate conv's first entry; gate conv entry spread %.2f us, exit (max) %.1f us after its first entry" % (
    wall[:, 8].min() - res[:, 8].min(), wall[:, 8].min() - res[:, 14].max(), wall[:, 8].max() - wall[:, 8].min(), wall[:, 14].max() - wall[:, 8].min()))
cyc = np.frombuffer(cb, dtype=np.uint64).reshape(512, 16).astype(np.float64)[:256]
inn = ["top wait", "top barrier", "block 0", "blocks 1-2", "mid wait", "mid barrier", "blocks 3-5"]
d = np.diff(cyc[:, :8], axis=1)
print("inside one chunk, cycles (median / mean / p90 over workgroups):")
for n, col in zip(inn, d.T):
    print("  %-12s %6.0f %6.0f %6.0f" % (n, np.median(col), col.mean(), np.percentile(col, 90)))
print("  %-12s %6.0f   (MFMA issue: 2304 per SIMD)" % ("chunk", np.median(cyc[:, 7] - cyc[:, 0])))
t0 = wall[:, 8].min()
names = ["entry", "loop", "ml0", "ep0", "ml1", "ep1", "exit"]
tl = wall[:, 8:15] - t0
print("timeline, us from the first workgroup's entry (mean / max): " + "  ".join("%s %.1f/%.1f" % (n, v, m) for n, v, m in zip(names, tl.mean(axis=0), tl.max(axis=0))))
dc, dw = np.diff(cyc[:, 8:15], axis=1), np.diff(wall[:, 8:15], axis=1) * 100.0
print("clock (GHz): " + "  ".join("%s %.2f" % (n, v) for n, v in zip(names[1:], np.median(dc / np.maximum(dw, 1) / 10.0, axis=0))))
print("cycles per chunk over tile 0's main loop: %.0f" % np.median((cyc[:, 10] - cyc[:, 9]) / 27.0))
