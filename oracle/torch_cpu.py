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


def noncausal_layer(Wg, Wv, Og, Ov, x, y, dilation, last_layer):
    """NonCausalLayer.forward on its own (model/waveglow.py:41-46): -> (x + res or None, skip).  Weights as (g or None, v) pairs."""
    W = _wn(None if Wg is None else torch.from_numpy(Wg), torch.from_numpy(Wv))
    Wo = _wn(None if Og is None else torch.from_numpy(Og), torch.from_numpy(Ov))
    xt = torch.from_numpy(np.ascontiguousarray(x, np.float32))
    radix, Cd, C = W.shape[2], W.shape[0] // 2, W.shape[1]
    xy = Fn.conv1d(xt, W, padding=dilation * (radix - 1) // 2, dilation=dilation) + torch.from_numpy(np.ascontiguousarray(y, np.float32))
    o = Fn.conv1d(torch.tanh(xy[:, :Cd]) * torch.sigmoid(xy[:, Cd:]), Wo)
    if last_layer:
        return None, o.numpy()
    return (o[:, :C] + xt).numpy(), o[:, C:].numpy()


def wn2d_forward(params, n_group, x, y):
    """WN2D.forward on its own (model/waveflow.py:70-135, layers :41-51): 3x3 convs dilated (h_dilation, 2^i), causal along the height axis
    (top padding only), the conditioning projection broadcast over the height axis.  params: [Vg, Vv, Sg, Sv, (Wg, Wv, Og, Ov) * 8, End] (+ the 19 biases, WN2D(bias=True))."""
    p = [None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)) for a in params]
    xt, yt = torch.from_numpy(np.ascontiguousarray(x, np.float32)), torch.from_numpy(np.ascontiguousarray(y, np.float32))
    with torch.no_grad():
        ls, t = wn2d_forward_t(p, n_group, xt, yt)
    return ls.numpy(), t.numpy()


def wn2d_forward_t(p, n_group, xt, yt):
    """wn2d_forward on torch tensors (differentiable: the checker of WN2D's gradients when the module is called on its own)."""
    hds = {8: [1] * 8, 16: [1] * 8, 32: [1, 2, 4] * 2 + [1, 2], 64: [1, 2, 4, 8, 16, 1, 2, 4], 128: [1, 2, 4, 8, 16, 32, 64, 1]}[n_group]
    b = p[37:] if len(p) > 37 else [None] * 19          # bias=True: V.bias, start.bias, (W.bias, W_o.bias) x 8, end.bias behind end.weight
    h = Fn.conv2d(xt, _wn(p[2], p[3]), b[1])
    v = Fn.conv1d(yt, _wn(p[0], p[1]), b[0]).unsqueeze(2)
    Cd = p[5].shape[0] // 2
    C = p[5].shape[1]
    skip = None
    for i in range(8):
        Wg, Wv, Og, Ov = p[4 + 4 * i: 8 + 4 * i]
        d, hd = 2 ** i, hds[i]
        xy = Fn.conv2d(Fn.pad(h, [d, d, 2 * hd, 0]), _wn(Wg, Wv), b[2 + 2 * i], dilation=(hd, d)) + v[:, 2 * Cd * i: 2 * Cd * (i + 1)]
        o = Fn.conv2d(torch.tanh(xy[:, :Cd]) * torch.sigmoid(xy[:, Cd:]), _wn(Og, Ov), b[3 + 2 * i])
        if i < 7:
            h = o[:, :C] + h
            sk = o[:, C:]
        else:
            sk = o
        skip = sk if skip is None else skip + sk
    out = Fn.conv2d(skip, p[36], b[18])
    return out[:, :1], out[:, 1:]


def waveflow_train_step(cfg, params, audio, mel, sigma, need_dh=False, double=False):
    """One training step of WaveFlow(use_conv1x1=False) (model/waveflow.py:196-223: replication pad + full ConvTranspose1d + LeakyReLU(0.4)
    upsampler :169-175; squeeze to [B, 1, n_group, W]; per flow (log_s, t) = WN2D(x[:, :, :-1], y), rows 1.. mapped affinely, the rows flipped
    with row 0 at the end :206-218) and the NLL of model/loss.py:10-15.  params in tests/golden/fill.waveflow_param_specs order (no 1x1 convs).
    The graph is held for ONE flow at a time: the forward keeps every flow's input (n_group x W values per item), the backward re-runs a
    flow under autograd from its input -- plain autograd in effect (the shipped config has memory_efficient=false), at an eighth of the memory.
    Returns dict(z, logdet, loss, grads, dh) like train_step; double: float64 arithmetic on the same float32 inputs."""
    flows, H, M = cfg["flows"], cfg["n_group"], cfg["n_mels"]
    dt = torch.float64 if double else torch.float32
    P = [None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dt) for a in params]
    per = 37 + (19 if cfg.get("bias") else 0)
    wn_p = [P[3 + k * per: 3 + (k + 1) * per] for k in range(flows)]
    x_in = torch.from_numpy(np.ascontiguousarray(audio, np.float32)).to(dt)
    hm = torch.from_numpy(np.ascontiguousarray(mel, np.float32)).to(dt)
    B, N = x_in.shape
    W = N // H
    s = 256 // H

    def upsample(bias, g, v, hh):
        y = Fn.conv_transpose1d(Fn.pad(hh, [0, 1], mode="replicate"), _wn(g, v), bias, stride=s, padding=s // 2)
        return Fn.leaky_relu(y, 0.4)[..., :W]

    def flow(p, x, y):
        log_s, t = wn2d_forward_t(p, H, x[:, :, :-1], y)
        xout = x[:, :, 1:] * torch.exp(log_s) + t
        return torch.cat((xout.flip(2), x[:, :, :1]), 2), log_s.sum((1, 2, 3))

    with torch.no_grad():
        y = upsample(P[0], P[1], P[2], hm)
        x = x_in.view(B, 1, W, H).transpose(2, 3).contiguous()
        logdet = torch.zeros(B, dtype=dt)
        xs = []
        for k in range(flows):
            xs.append(x)
            x, ld = flow(wn_p[k], x, y)
            logdet = logdet + ld
        z = x.squeeze(1).transpose(1, 2).contiguous().view(B, N)
        loss = (0.5 * (z * z).sum(1) / sigma ** 2 - logdet).mean() / N
        dx = (z / (sigma ** 2 * B * N)).view(B, 1, W, H).transpose(2, 3).contiguous()
    dld = torch.full((B,), -1.0 / (B * N), dtype=dt)
    grads = [None if p is None else torch.zeros_like(p) for p in P]
    dy = torch.zeros_like(y)
    for k in range(flows - 1, -1, -1):
        with torch.enable_grad():
            xi = xs[k].detach().requires_grad_(True)
            y_ = y.detach().requires_grad_(True)
            live = [q.detach().requires_grad_(True) if q is not None else None for q in wn_p[k]]
            xo, ld = flow(live, xi, y_)
            wanted = [xi, y_] + [q for q in live if q is not None]
            got = torch.autograd.grad([xo, ld], wanted, [dx, dld])
        dx = got[0]
        dy += got[1]
        it = iter(got[2:])
        for j, q in enumerate(live):
            if q is not None:
                grads[3 + k * per + j] = next(it)
    with torch.enable_grad():
        live = [q.detach().requires_grad_(True) if q is not None else None for q in P[:3]]
        h_ = hm.detach().requires_grad_(need_dh)
        yy = upsample(live[0], live[1], live[2], h_)
        wanted = [q for q in live if q is not None] + ([h_] if need_dh else [])
        got = torch.autograd.grad(yy, wanted, dy)
    it = iter(got)
    for j, q in enumerate(live):
        if q is not None:
            grads[j] = next(it)
    dh = next(it).numpy() if need_dh else None
    return dict(z=z.numpy(), logdet=logdet.numpy(), loss=float(loss), grads=[None if g is None else g.numpy() for g in grads], dh=dh)


def _step_of(cfg):
    """the training step a job's cfg asks for: cfg['model'] == 'waveflow' -> waveflow_train_step, else train_step (WaveGlow)"""
    return waveflow_train_step if cfg.get("model") == "waveflow" else train_step


def set_threads(n):
    torch.set_num_threads(max(1, int(n)))
    return torch.get_num_threads()


@torch.no_grad()
def train_step(cfg, params, audio, h, sigma, need_dh=False, double=False):
    """cfg: dict of the reference's WaveGlow ctor keywords.  Returns dict(z, logdet, loss, grads [list like params], dh).
    double: float64 arithmetic on the same float32 inputs (the checker for the headline shape; ATen's plain convolution path)."""
    flows, G, depth, radix = cfg["flows"], cfg["n_group"], cfg.get("depth", 8), cfg.get("radix", 3)
    C = cfg.get("residual_channels", 256)
    dt = torch.float64 if double else torch.float32
    P = [None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dt) for a in params]
    x_in = torch.from_numpy(np.ascontiguousarray(audio, np.float32)).to(dt)
    hm = torch.from_numpy(np.ascontiguousarray(h, np.float32)).to(dt)
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
    logdet = torch.zeros(B, dtype=dt)
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


# ------------------------------------------------------------------------------------------------------------------------------
# Batch items are independent units, so a step over B segments is W worker PROCESSES with a share of the batch each:
# ATen's intra-op pool does not scale on these shapes (T = 2000 columns per convolution), W processes x a few threads do.  Used by
#   * tests/test_gpu_parity.py::test_c2_full_batch_vs_oracle -- the B = 24 x 16000 headline shape against this oracle;
#   * bench.py's cpu_baseline -- the all-core CPU figure (`time_parallel`).
# A worker is `python -m oracle.torch_cpu --worker <dir> <index>`: inputs and outputs travel as .npy / .npz files in a scratch directory.
# Combination: with the loss normalised by B N, the full-batch gradient is sum_w (B_w / B) grad_w -- including the 1x1 convs'
# W^-T (sum_b dlogdet_b) T term, which is -T / N in every share -- and the loss the same weighted mean.
# ------------------------------------------------------------------------------------------------------------------------------
def _scratch():
    import tempfile
    base = "/dev/shm" if __import__("os").path.isdir("/dev/shm") else None
    return tempfile.mkdtemp(prefix="wg_torch_cpu_", dir=base)


def host_cpu_budget():
    """What this process may really use: (logical CPUs in its affinity mask, CPUs' worth of cgroup quota or None, physical cores among the
    allowed CPUs as a list of one logical CPU per core).  A container may show 128 cores and grant a fraction of them."""
    import os
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:                                            # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:                                        # noqa: BLE001
            pass
    firsts, seen = [], set()
    for c in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except Exception:                                        # noqa: BLE001
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            firsts.append(c)
    return len(allowed), quota, firsts


def _spawn(workdir, n):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="", MKL_NUM_THREADS="")
    env.pop("OMP_NUM_THREADS"); env.pop("MKL_NUM_THREADS")
    return [subprocess.Popen([sys.executable, "-m", "oracle.torch_cpu", "--worker", workdir, str(i)], env=env, cwd=root) for i in range(n)]


def _write_job(workdir, cfg, params, audio, h, sigma, shares, threads, double, need_dh, runs):
    import json
    import os
    np.savez(os.path.join(workdir, "params.npz"), **{"p%d" % i: a for i, a in enumerate(params) if a is not None})
    np.save(os.path.join(workdir, "audio.npy"), np.ascontiguousarray(audio, np.float32))
    np.save(os.path.join(workdir, "h.npy"), np.ascontiguousarray(h, np.float32))
    with open(os.path.join(workdir, "job.json"), "w") as f:
        # every worker gets physical cores of its own (one logical CPU per core, consecutive cores): without it the workers' thread
        # pools were free to pile onto the same cores
        _, _, firsts = host_cpu_budget()
        cpus = [firsts[i * int(threads):(i + 1) * int(threads)] for i in range(len(shares))]
        if any(len(c) < int(threads) for c in cpus):
            cpus = None
        json.dump(dict(cfg=cfg, n_params=len(params), none=[i for i, a in enumerate(params) if a is None], sigma=float(sigma),
                       shares=shares, threads=int(threads), double=bool(double), need_dh=bool(need_dh), runs=int(runs), cpus=cpus), f)


def _wait(procs, what):
    for p in procs:
        if p.wait() != 0:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            raise RuntimeError("oracle/torch_cpu.py: a worker process of %s failed (exit code %s)" % (what, p.returncode))


def _shares(B, workers):
    workers = max(1, min(int(workers), B))
    cut = [round(i * B / workers) for i in range(workers + 1)]
    return [(cut[i], cut[i + 1]) for i in range(workers) if cut[i + 1] > cut[i]]


def train_step_parallel(cfg, params, audio, h, sigma, workers, threads=8, need_dh=False, double=False):
    """train_step (or, with cfg['model'] == 'waveflow', waveflow_train_step) over the batch cut into `workers` shares, one process each
    (see above).  Same return value as train_step (arrays in float64 when double)."""
    import os
    import shutil
    B, N = audio.shape
    shares = _shares(B, workers)
    workdir = _scratch()
    try:
        _write_job(workdir, cfg, params, audio, h, sigma, shares, threads, double, need_dh, 0)
        _wait(_spawn(workdir, len(shares)), "train_step_parallel")
        out = None
        for i, (b0, b1) in enumerate(shares):
            r = np.load(os.path.join(workdir, "out%d.npz" % i))
            w = (b1 - b0) / B
            if out is None:
                out = dict(z=np.empty((B, N), r["z"].dtype), logdet=np.empty(B, r["logdet"].dtype), loss=0.0,
                           grads=[None if ("g%d" % j) not in r else np.zeros(r["g%d" % j].shape, np.float64) for j in range(len(params))],
                           dh=np.empty(h.shape, r["dh"].dtype) if need_dh else None)
            out["z"][b0:b1], out["logdet"][b0:b1] = r["z"], r["logdet"]
            out["loss"] += w * float(r["loss"])
            for j, g in enumerate(out["grads"]):
                if g is not None:
                    g += w * r["g%d" % j]
            if need_dh:
                out["dh"][b0:b1] = w * r["dh"]                  # d loss / d h_b carries the 1 / (B N) of ITS share's loss
        if not double:
            out["grads"] = [None if g is None else g.astype(np.float32) for g in out["grads"]]
        return out
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


def time_parallel(cfg, params, audio, h, sigma, workers, threads=8, runs=3):
    """All-core throughput: `workers` processes, each with ONE segment (row i % B of `audio`), 1 warm-up step, then -- all released
    together -- `runs` timed steps.  Returns dict(samples_per_s = workers * runs * N / (last end - first start), per-worker medians)."""
    import json
    import os
    import shutil
    import time
    B, N = audio.shape
    shares = [(i % B, i % B + 1) for i in range(int(workers))]
    workdir = _scratch()
    try:
        _write_job(workdir, cfg, params, audio, h, sigma, shares, threads, False, False, runs)
        procs = _spawn(workdir, len(shares))
        t_wait = time.time()
        while sum(os.path.exists(os.path.join(workdir, "ready%d" % i)) for i in range(len(shares))) < len(shares):
            if any(p.poll() not in (None, 0) for p in procs) or time.time() - t_wait > 900:
                for q in procs:
                    if q.poll() is None:
                        q.kill()
                raise RuntimeError("oracle/torch_cpu.py: a timing worker failed or did not come up")
            time.sleep(0.05)
        open(os.path.join(workdir, "go"), "w").close()
        _wait(procs, "time_parallel")
        t = [json.load(open(os.path.join(workdir, "time%d.json" % i))) for i in range(len(shares))]
        start, end = min(x["start"] for x in t), max(x["end"] for x in t)
        med = sorted(sorted(x["steps"])[len(x["steps"]) // 2] for x in t)
        return {"samples_per_s": len(shares) * runs * N / (end - start), "wall_s": end - start, "workers": len(shares), "threads_per_worker": int(threads),
                "runs": int(runs), "step_s_median_fastest_worker": med[0], "step_s_median_slowest_worker": med[-1]}
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


def _worker_main(workdir, index):
    import json
    import os
    import time
    job = json.load(open(os.path.join(workdir, "job.json")))
    if job.get("cpus"):
        try:
            os.sched_setaffinity(0, job["cpus"][index])
        except Exception:                                        # noqa: BLE001 -- a refused mask only costs speed
            pass
    set_threads(job["threads"])
    pz = np.load(os.path.join(workdir, "params.npz"))
    params = [None if i in set(job["none"]) else pz["p%d" % i] for i in range(job["n_params"])]
    b0, b1 = job["shares"][index]
    audio = np.array(np.load(os.path.join(workdir, "audio.npy"), mmap_mode="r")[b0:b1])      # (copies: torch wants writable arrays)
    h = np.array(np.load(os.path.join(workdir, "h.npy"), mmap_mode="r")[b0:b1])
    train_step_ = _step_of(job["cfg"])
    if job["runs"]:                                          # timing worker
        train_step_(job["cfg"], params, audio, h, job["sigma"])
        open(os.path.join(workdir, "ready%d" % index), "w").close()
        while not os.path.exists(os.path.join(workdir, "go")):
            time.sleep(0.005)
        steps, start = [], time.time()
        for _ in range(job["runs"]):
            t0 = time.time()
            train_step_(job["cfg"], params, audio, h, job["sigma"])
            steps.append(time.time() - t0)
        with open(os.path.join(workdir, "time%d.json" % index), "w") as f:
            json.dump({"start": start, "end": time.time(), "steps": steps}, f)
        return
    r = train_step_(job["cfg"], params, audio, h, job["sigma"], need_dh=job["need_dh"], double=job["double"])
    out = {"z": r["z"], "logdet": r["logdet"], "loss": np.float64(r["loss"])}
    for j, g in enumerate(r["grads"]):
        if g is not None:
            out["g%d" % j] = g
    if job["need_dh"]:
        out["dh"] = r["dh"]
    np.savez(os.path.join(workdir, "out%d.npz" % index), **out)


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 4 and sys.argv[1] == "--worker":
        _worker_main(sys.argv[2], int(sys.argv[3]))
    else:
        raise SystemExit("usage: python -m oracle.torch_cpu --worker <scratch dir> <index>   (started by train_step_parallel / time_parallel)")
