"""torch-CPU restatement of the WaveGlow training step  (TEST INFRASTRUCTURE -- the checker / CPU baseline, never the product).

The same algorithm as oracle/wg_oracle.c, written on the library the reference itself runs on (ATen / MKLDNN convolutions), so that the
CPU baseline timed next to the GPU number moves at the speed of the reference's own CPU path instead of a plain-C loop nest's.  Own code:
nothing is imported from the reference; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module.

What it follows (paths relative to the upstream repository):
  * upsampler: depthwise ConvTranspose1d + old-style weight norm (dim 0), cropped to T        model/waveglow.py:126-130,151,156-157
  * squeeze / early outputs / unsqueeze                                                          model/waveglow.py:153,164-170,178-179
  * InvertibleConv1x1: z = W x, logdet += T log det W (a scalar shared by the batch)             model/efficient_modules.py:37-41,219-223
  * WN: start, V (chunked per layer), dilated k=3 convs, gate, W_o -> (res, skip), end -> (log_s, t)   model/waveglow.py:41-46,98-105
  * coupling: zb = xb exp(log_s) + t                                                            model/efficient_modules.py:105-111
  * NLL: mean_b(0.5 sum z^2 / sigma^2 - logdet) / N                                              model/loss.py:10-15
  * constant-memory backward: flows last to first; each flow's WN is recomputed WITH a local autograd graph from the flow's output, the
    flow input is rebuilt (xb = (zb - t) / s, x = W^-1 z), one autograd.grad of cat(log_s, t) w.r.t. [xa, y, parameters] with
    grad_outputs = cat(dzb xb s + dlog_s, dzb); dW = sum dz x^T + W^-T (sum_b dlogdet_b) T              model/efficient_modules.py:118-154,230-244
No autograd graph ever spans more than one WN: activation memory is O(1) in the number of flows, as in the reference.

Pinned to the reference's golden fixtures by tests/test_oracle_golden.py (z <= 1e-6, gradients <= 1e-5 of each tensor's max).
Parameter table: a list of float32 numpy arrays in the order of tests/golden/fill.model_param_specs (= the reference's named_parameters()).
"""
import numpy as np
import torch
import torch.nn.functional as Fn


def _wn(g, v):
    """old-style weight norm over all dims but 0 (utils.py:14-16): w = g v / ||v||; g None -> plain weight"""
    if g is None:
        return v
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


def _flow_channels(cfg, k):
    return cfg["n_group"] - cfg["n_early_size"] * (k // cfg["n_early_every"])


def _wn_forward(p, xa, y, depth, C, radix):
    """p: this WN's tensors [Vg, Vv, Sg, Sv, (Wg, Wv, Og, Ov) * depth, End] and, for WN(bias=True) (model/waveglow.py:58), the biases
    behind them [V, start, (W, W_o) * depth, end].  -> (log_s, t)"""
    nb = 4 + 4 * depth + 1
    b = p[nb:] if len(p) > nb else [None] * (2 + 2 * depth + 1)
    x = Fn.conv1d(xa, _wn(p[2], p[3]), b[1])
    v = Fn.conv1d(y, _wn(p[0], p[1]), b[0])
    Cd = p[5].shape[0] // 2
    skip = None
    for i in range(depth):
        Wg, Wv, Og, Ov = p[4 + 4 * i: 8 + 4 * i]
        d = 2 ** i
        xy = Fn.conv1d(x, _wn(Wg, Wv), b[2 + 2 * i], padding=d * (radix - 1) // 2, dilation=d) + v[:, 2 * Cd * i: 2 * Cd * (i + 1)]
        gate = torch.tanh(xy[:, :Cd]) * torch.sigmoid(xy[:, Cd:])
        o = Fn.conv1d(gate, _wn(Og, Ov), b[3 + 2 * i])
        if i < depth - 1:
            x = o[:, :C] + x
            s = o[:, C:]
        else:
            s = o
        skip = s if skip is None else skip + s
    out = Fn.conv1d(skip, p[nb - 1], b[2 + 2 * depth])
    ic = out.shape[1] // 2
    return out[:, :ic], out[:, ic:]


def set_threads(n):
    torch.set_num_threads(max(1, int(n)))
    return torch.get_num_threads()


@torch.no_grad()
def train_step(cfg, params, audio, h, sigma, need_dh=False):
    """cfg: dict of the reference's WaveGlow ctor keywords.  Returns dict(z, logdet, loss, grads [list like params], dh)."""
    flows, G, depth, radix = cfg["flows"], cfg["n_group"], cfg.get("depth", 8), cfg.get("radix", 3)
    C = cfg.get("residual_channels", 256)
    P = [None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in params]
    x_in, hm = torch.from_numpy(np.ascontiguousarray(audio, np.float32)), torch.from_numpy(np.ascontiguousarray(h, np.float32))
    B, N = x_in.shape
    T = N // G
    up = cfg["hop_size"] // G
    K = 2 * up + 1
    pad = K // 2 - up // 2
    per_wn = 4 + 4 * depth + 1 + (2 + 2 * depth + 1 if cfg.get("bias") else 0)
    wn_p = [P[3 + flows + k * per_wn: 3 + flows + (k + 1) * per_wn] for k in range(flows)]

    def upsample(bias, g, v, hh):
        return Fn.conv_transpose1d(hh, _wn(g, v), bias, stride=up, padding=pad, groups=cfg["n_mels"])[..., :T]

    y = upsample(P[0], P[1], P[2], hm)
    x = x_in.view(B, T, G).transpose(1, 2).contiguous()
    logdet = torch.zeros(B)
    early, outs = [], []                                   # emitted channels; per-flow outputs (only views of what the backward needs)
    for k in range(flows):
        if k and k % cfg["n_early_every"] == 0:
            early.append(x[:, :cfg["n_early_size"]])
            x = x[:, cfg["n_early_size"]:]
        c = x.shape[1]
        W = P[3 + k].view(c, c)
        x = torch.matmul(W, x)
        logdet = logdet + T * torch.logdet(W)
        log_s, t = _wn_forward(wn_p[k], x[:, :c // 2], y, depth, C, radix)
        x = torch.cat((x[:, :c // 2], x[:, c // 2:] * torch.exp(log_s) + t), 1)
        logdet = logdet + log_s.sum((1, 2))
    Z = torch.cat(early + [x], 1)
    z = Z.transpose(1, 2).contiguous().view(B, N)
    loss = (0.5 * (z * z).sum(1) / sigma ** 2 - logdet).mean() / N

    # ---- backward: d loss / d z = z / (sigma^2 B N), d loss / d logdet_b = -1 / (B N) ------------------------------------------
    grads = [None if p is None else torch.zeros_like(p) for p in P]
    dZ = Z / (sigma ** 2 * B * N)
    dld = -1.0 / (B * N)
    n_early = sum(e.shape[1] for e in early)
    zc, dzc = Z[:, n_early:], dZ[:, n_early:]
    dy = torch.zeros_like(y)
    ei = len(early)
    for k in range(flows - 1, -1, -1):
        c = zc.shape[1]
        ic = c // 2
        za, zb, dza, dzb = zc[:, :ic], zc[:, ic:], dzc[:, :ic], dzc[:, ic:]
        with torch.enable_grad():
            xa_ = za.detach().requires_grad_(True)
            y_ = y.detach().requires_grad_(True)
            live = [p.detach().requires_grad_(True) if p is not None else None for p in wn_p[k]]
            log_s, t = _wn_forward(live, xa_, y_, depth, C, radix)
            s = torch.exp(log_s.detach())
            xb = (zb - t.detach()) / s
            wanted = [xa_, y_] + [p for p in live if p is not None]
            got = torch.autograd.grad(torch.cat((log_s, t), 1), wanted, torch.cat((dzb * xb * s + dld, dzb), 1))
        dxa = dza + got[0]
        dy += got[1]
        it = iter(got[2:])
        base = 3 + flows + k * per_wn
        for j, p in enumerate(live):
            if p is not None:
                grads[base + j] = next(it)
        xk = torch.cat((za, xb), 1)                                    # the coupling's input = the 1x1's output
        dxk = torch.cat((dxa, dzb * s), 1)
        W = P[3 + k].view(c, c)
        Winv = torch.inverse(W)
        xin = torch.matmul(Winv, xk)                                   # rebuilt flow input
        grads[3 + k] = (torch.einsum("bit,bjt->ij", dxk, xin) + Winv.t() * (dld * B * T)).view(c, c, 1)
        dxin = torch.matmul(W.t(), dxk)
        if k and k % cfg["n_early_every"] == 0:                        # re-attach the channels emitted in front of this flow
            ei -= 1
            ne = early[ei].shape[1]
            n_early -= ne
            zc = torch.cat((Z[:, n_early:n_early + ne], xin), 1)
            dzc = torch.cat((dZ[:, n_early:n_early + ne], dxin), 1)
        else:
            zc, dzc = xin, dxin
    with torch.enable_grad():
        live = [p.detach().requires_grad_(True) if p is not None else None for p in P[:3]]
        h_ = hm.detach().requires_grad_(need_dh)
        yy = upsample(live[0], live[1], live[2], h_)
        wanted = [p for p in live if p is not None] + ([h_] if need_dh else [])
        got = torch.autograd.grad(yy, wanted, dy)
    it = iter(got)
    for j, p in enumerate(live):
        if p is not None:
            grads[j] = next(it)
    dh = next(it).numpy() if need_dh else None
    return dict(z=z.numpy(), logdet=logdet.numpy(), loss=float(loss), grads=[None if g is None else g.numpy() for g in grads], dh=dh)
