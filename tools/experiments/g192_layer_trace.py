"""Phase stamps of convlayer16g_kernel (developer experiment; -DWG_DBG_TRACE build): the gate product's and the residual product's
timeline inside ONE launch, microseconds from the first workgroup's entry (mean / max over the 256 workgroups).

    WGFLOW_LIB=variants/lib_gltrace.so python tools/experiments/g192_layer_trace.py"""
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
    for _ in range(20):
        blk(x.clone(), y)
torch.cuda.synchronize()
L = _lib.lib()
N = 512 * 16
wb = (C.c_ulonglong * N)()
L.wg_dbg_trace_read.argtypes = [C.c_void_p, C.c_int]
assert L.wg_dbg_trace_read(wb, N) == 0
wall = np.frombuffer(wb, dtype=np.uint64).reshape(512, 16).astype(np.float64) / 100.0
G, R = wall[:256], wall[256:]
t0 = G[:, 8].min()
for name, T, slots in (("gate product", G, (8, 9, 10, 11, 12, 13, 14)), ("residual product", R, (8, 9, 10, 11, 14))):
    names = {8: "entry", 9: "loop", 10: "ml0", 11: "ep0", 12: "ml1", 13: "ep1", 14: "exit"}
    print("%-17s" % name + "  ".join("%s %.1f/%.1f" % (names[s], (T[:, s] - t0).mean(), (T[:, s] - t0).max()) for s in slots))
