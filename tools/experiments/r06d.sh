cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "test_wide_batch_step_vs_oracle and bf16x3p" 2>&1 | grep -E "assert|Error|error|relmax|^E " | head -30
python - <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden'); sys.path.insert(0, '.')
import fill
from oracle import wg_oracle as orc
import constant_memory_waveglow_amd as cm
import importlib
tg = importlib.import_module('test_gpu_parity')
dev = torch.device('cuda:0')
os.environ['WG_PRECISION'] = 'bf16x3p'
res = {}
for lr in ('0', '1'):
    os.environ['WG_LOWRANK'] = lr
    cm._lib.lib().wg_reload_env()
    m, cfg, specs, P = tg.build("c1", dev)
    B, (_, N, F) = 9, fill.SHAPES["c1"]
    audio, h = fill.inputs("c1x9", B, N, F, cfg["n_mels"])
    x, ht = tg.T(audio, dev), tg.T(h, dev).requires_grad_(True)
    z, logdet = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    res[lr] = (z.detach().cpu().numpy(), logdet.detach().cpu().numpy(), float(loss), {n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters()}, ht.grad.cpu().numpy())
a, b = res['0'], res['1']
print('z', np.abs(a[0] - b[0]).max(), 'logdet', np.abs(a[1] - b[1]).max(), 'loss', a[2], b[2], 'dh', np.abs(a[4]-b[4]).max()/np.abs(a[4]).max())
for n in a[3]:
    e = np.abs(a[3][n] - b[3][n]).max() / max(np.abs(a[3][n]).max(), 1e-30)
    if e > 1e-5: print(n, a[3][n].shape, e)
PY
