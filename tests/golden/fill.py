"""Deterministic synthetic tensors for fixtures, parity tests and the smoke check.

Every tensor is a pure function of (its name, its shape, a kind) through a counter-based generator
(splitmix64 of a name hash + element index), so fixtures only need to store OUTPUTS: inputs and
weights are regenerated bit-for-bit wherever the tests run.  No torch, no global RNG state.
"""
import hashlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _key(name):
    return np.uint64(int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little"))


def uniform01(name, n):
    """n float64 in (0,1), 24 significant bits, keyed by name."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(_splitmix64(idx + _key(name)))
    return ((bits >> np.uint64(40)).astype(np.float64) + 0.5) / float(1 << 24)


def uniform(name, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(name, n)).astype(np.float32).reshape(shape)


def normal(name, shape, std=1.0):
    n = int(np.prod(shape))
    u1 = uniform01(name + "#u1", n)
    u2 = uniform01(name + "#u2", n)
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (std * g).astype(np.float32).reshape(shape)


def orthogonal(name, c):
    """c x c orthogonal matrix with det > 0 (modified Gram-Schmidt of a keyed gaussian matrix);
    plays the role of the reference's QR init (model/efficient_modules.py:22-26), perturbed a little so
    that W is well conditioned but NOT orthogonal (logdet != 0, W^-1 != W^T)."""
    a = normal(name, (c, c)).astype(np.float64)
    q = np.zeros_like(a)
    for j in range(c):
        v = a[:, j].copy()
        for i in range(j):
            v = v - (q[:, i] * v).sum() * q[:, i]
        q[:, j] = v / np.sqrt((v * v).sum())
    q = q + 0.15 * normal(name + "#pert", (c, c)).astype(np.float64) / np.sqrt(c)
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.astype(np.float32)


def flow_channels(cfg, k):
    c = cfg["n_group"]
    for j in range(1, k + 1):
        if j % cfg["n_early_every"] == 0:
            c -= cfg["n_early_size"]
    return c


def wn_param_specs(prefix, in_ch, aux, C, Cd, Cs, depth, radix, bias=False):
    """(name, shape, kind) of one WN in the reference's parameter order (model/waveglow.py:70-96).  bias=True (waveglow.py:58): the
    biases follow behind `end.weight` -- V, start, per layer (W, W_o), end -- which is the order of the C ABI's / the oracle's table
    (the reference's own named_parameters() interleaves them; state dicts go by name, so only the table order is ours)."""
    specs = [(prefix + "V.weight_g", (2 * Cd * depth, 1, 1), "g"), (prefix + "V.weight_v", (2 * Cd * depth, aux, 1), "v"),
             (prefix + "start.weight_g", (C, 1, 1), "g"), (prefix + "start.weight_v", (C, in_ch, 1), "v")]
    for i in range(depth):
        rows = Cs if i == depth - 1 else C + Cs
        specs += [(prefix + "layers.%d.W.weight_g" % i, (2 * Cd, 1, 1), "g"),
                  (prefix + "layers.%d.W.weight_v" % i, (2 * Cd, C, radix), "v"),
                  (prefix + "layers.%d.W_o.weight_g" % i, (rows, 1, 1), "g"),
                  (prefix + "layers.%d.W_o.weight_v" % i, (rows, Cd, 1), "v")]
    specs.append((prefix + "end.weight", (2 * in_ch, Cs, 1), "end"))
    if bias:
        specs += [(prefix + "V.bias", (2 * Cd * depth,), "bias"), (prefix + "start.bias", (C,), "bias")]
        for i in range(depth):
            rows = Cs if i == depth - 1 else C + Cs
            specs += [(prefix + "layers.%d.W.bias" % i, (2 * Cd,), "bias"), (prefix + "layers.%d.W_o.bias" % i, (rows,), "bias")]
        specs.append((prefix + "end.bias", (2 * in_ch,), "bias"))
    return specs


def model_param_specs(cfg):
    """cfg: dict with the reference's WaveGlow ctor keywords (configs/waveglow_LJ_speech.json:6-19)."""
    up = cfg["hop_size"] // cfg["n_group"]
    K = 2 * up + 1
    M = cfg["n_mels"]
    specs = [("upsampler.bias", (M,), "bias"), ("upsampler.weight_g", (M, 1, 1), "g"), ("upsampler.weight_v", (M, 1, K), "v")]
    for k in range(cfg["flows"]):
        c = flow_channels(cfg, k)
        specs.append(("invconv1x1.%d.weight" % k, (c, c, 1), "orth"))
    for k in range(cfg["flows"]):
        c = flow_channels(cfg, k)
        specs += wn_param_specs("WNs.%d.F." % k, c // 2, M, cfg["residual_channels"], cfg["dilation_channels"],
                                cfg["skip_channels"], cfg["depth"], cfg["radix"], bias=cfg.get("bias", False))
    return specs


def fill_params(specs, tag=""):
    """dict name -> float32 array.  `v` ~ U(+-1/sqrt(fan_in)); `g` = ||v|| * (1 + 0.2 u) (so g != ||v||);
    `end` ~ N(0, (0.25/sqrt(Cs))^2) (NOT the reference's zero init: log_s, t must be non-trivial)."""
    out = {}
    for name, shape, kind in specs:
        key = tag + name
        if kind == "v":
            fan_in = int(np.prod(shape[1:]))
            b = 1.0 / np.sqrt(fan_in)
            out[name] = uniform(key, shape, -b, b)
        elif kind == "bias":
            out[name] = uniform(key, shape, -0.1, 0.1)
        elif kind == "end":
            out[name] = normal(key, shape, 0.25 / np.sqrt(shape[1]))
        elif kind == "orth":
            out[name] = orthogonal(key, shape[0]).reshape(shape)
    for name, shape, kind in specs:
        if kind == "g":
            v = out[name[:-1] + "v"].astype(np.float64)
            nrm = np.sqrt((v.reshape(shape[0], -1) ** 2).sum(1))
            out[name] = (nrm * (1.0 + 0.2 * uniform(tag + name, (shape[0],)).astype(np.float64))).astype(np.float32).reshape(shape)
    return out


def table(specs, params):
    """list of arrays in table (= named_parameters) order"""
    return [params[name] for name, _, _ in specs]


# named configurations shared by the fixture generator and the tests ---------------------------------
CONFIGS = {
    # tiny model whose FULL gradient set fits in a fixture
    "micro": dict(flows=4, n_group=8, n_early_every=2, n_early_size=2, hop_size=64, n_mels=20,
                  dilation_channels=32, residual_channels=32, skip_channels=32, depth=3, radix=3),
    # BASELINE.json configs[0]: 64ch, 6 flows
    "c1": dict(flows=6, n_group=8, n_early_every=4, n_early_size=2, hop_size=256, n_mels=80,
               dilation_channels=64, residual_channels=64, skip_channels=64, depth=8, radix=3),
    # BASELINE.json configs[1] (waveglow_LJ_speech.json): 256ch, 12 flows
    "c2": dict(flows=12, n_group=8, n_early_every=4, n_early_size=2, hop_size=256, n_mels=80,
               dilation_channels=256, residual_channels=256, skip_channels=256, depth=8, radix=3),
}
# WN(bias=True) (model/waveglow.py:58; no shipped config sets it): the micro model with a bias on every conv of every WN
CONFIGS["micro_bias"] = dict(CONFIGS["micro"], bias=True)
# WN(radix=5) (model/waveglow.py:57,28-30: any odd kernel size; the shipped configs use 3): five taps per dilated conv
CONFIGS["micro_r5"] = dict(CONFIGS["micro"], radix=5)
# the WaveGlow core of WSRGlow (model/wsrglow.py:22-25: n_group = hop = 8r, stride-1 upsampler, very wide conditioning), scaled
# down: odd conditioning width (not a multiple of 8), 16 squeezed channels, upsample factor 1
CONFIGS["wsr_like"] = dict(flows=4, n_group=16, n_early_every=2, n_early_size=2, hop_size=16, n_mels=83,
                           dilation_channels=64, residual_channels=64, skip_channels=64, depth=3, radix=3)
# WSRGlow(upsample_rate=2, **WSR_KW): the WaveGlow underneath is fixed by model/wsrglow.py:23-26
WSR_KW = dict(dilation_channels=32, residual_channels=32, skip_channels=32, depth=2, radix=3)
CONFIGS["wsr"] = dict(flows=12, n_group=16, n_early_every=4, n_early_size=2, hop_size=16, n_mels=8 * 400 + 51 * 9, **WSR_KW)
# WSRGlow(upsample_rate=3, **WSR_KW) (configs/wsrglow_vctk_3x.json): 24 squeezed channels -> 1x1 convs of 24, 22 and 20 channels
CONFIGS["wsr3"] = dict(flows=12, n_group=24, n_early_every=4, n_early_size=2, hop_size=24, n_mels=8 * 400 + 51 * 9, **WSR_KW)
# WSRGlow(upsample_rate=2) at the shipped width (configs/wsrglow_vctk_2x.json: WN defaults 256 / 256 / 256, depth 8; 229.7 M parameters) and
# its batch (12 x 8192): the timed workload of `bench.py --model wsrglow`, kept as a SUMMARY fixture (model_wsr_full.npz)
WSR_KW_FULL = dict(dilation_channels=256, residual_channels=256, skip_channels=256, depth=8, radix=3)
CONFIGS["wsr_full"] = dict(flows=12, n_group=16, n_early_every=4, n_early_size=2, hop_size=16, n_mels=8 * 400 + 51 * 9, **WSR_KW_FULL)
WSR_TABLES = [("mu_enc.1.weight", (256, 400)), ("angle_embed.embed.weight", (120, 50))]
SHAPES = {  # (batch, samples, mel frames)
    "micro": (2, 512, 8),
    "micro_bias": (2, 512, 8),
    "micro_r5": (2, 512, 8),
    "c1": (2, 4000, 16),
    "c2": (1, 16000, 63),
    "wsr_like": (2, 16 * 300, 300),
    "wsr": (2, 1024, 64),        # conditioning signal: [2, 512] low-rate samples -> 64 frames
    "wsr3": (2, 24 * 37, 37),    # rate 3: [2, 296] low-rate samples -> 37 frames (not a multiple of anything the kernels tile by)
}
SHAPES["wsr_full"] = (12, 8192, 512)
WSR_RATE = {"wsr": 2, "wsr3": 3, "wsr_full": 2}
# MelSpec cases (batch, samples): the conditioner every WaveGlow / WaveFlow config ships (sr 22050, n_fft 1024, hop 256, f_max 8000, 80 mels)
MEL_CASES = {"short": (2, 4096), "segment": (1, 16000)}
MEL_KW = dict(sr=22050, n_fft=1024, hop_length=256, f_max=8000, n_mels=80)
SIGMA = 0.7   # configs/waveglow_LJ_speech.json:47


def inputs(tag, B, N, F, n_mels):
    return uniform(tag + "/audio", (B, N), -1.0, 1.0), normal(tag + "/mel", (B, n_mels, F))


def wsr_inputs(tag, B, N, rate=2):
    """(audio [B,N], low-rate conditioning signal c [B,N/rate] in (-1.15, 1.15): exercises the clip and every mu-law level)."""
    return uniform(tag + "/audio", (B, N), -1.0, 1.0), uniform(tag + "/lowres", (B, N // rate), -1.15, 1.15)


def mel_input(tag):
    B, N = MEL_CASES[tag]
    x = uniform("mel/" + tag, (B, N), -0.8, 0.8)
    x[0, : N // 4] *= 1e-3                       # a quiet stretch: log-mel near its floor
    return x


def wsr_tables(tag):
    """the two embedding tables, N(0,1) like nn.Embedding's default init"""
    return {n: normal(tag + n, shp) for n, shp in WSR_TABLES}


# ---- WaveFlow (model/waveflow.py) ------------------------------------------------------------------

def waveflow_param_specs(cfg):
    """(name, shape, kind) in the reference's named_parameters() order for WaveFlow(use_conv1x1=False, bias=False):
    cfg has the ctor keywords of configs/waveflow_LJ_speech.json (flows, n_group, n_mels, *_channels)."""
    M, H = cfg["n_mels"], cfg["n_group"]
    s = 256 // H                                           # FlowBase hop is fixed to 256 (waveflow.py:160-163)
    C, Cd, Cs = cfg["residual_channels"], cfg["dilation_channels"], cfg["skip_channels"]
    specs = [("upsampler.1.bias", (M,), "bias"), ("upsampler.1.weight_g", (M, 1, 1), "g"), ("upsampler.1.weight_v", (M, M, 2 * s + 1), "v")]
    for k in range(cfg["flows"]):
        p = "WNs.%d." % k
        specs += [(p + "V.weight_g", (16 * Cd, 1, 1), "g"), (p + "V.weight_v", (16 * Cd, M, 1), "v"),
                  (p + "start.weight_g", (C, 1, 1, 1), "g"), (p + "start.weight_v", (C, 1, 1, 1), "v")]
        for i in range(8):
            rows = Cs if i == 7 else C + Cs
            specs += [(p + "layers.%d.W.weight_g" % i, (2 * Cd, 1, 1, 1), "g"), (p + "layers.%d.W.weight_v" % i, (2 * Cd, C, 3, 3), "v"),
                      (p + "layers.%d.W_o.weight_g" % i, (rows, 1, 1, 1), "g"), (p + "layers.%d.W_o.weight_v" % i, (rows, Cd, 1, 1), "v")]
        specs.append((p + "end.weight", (2, Cs, 1, 1), "end"))
        if cfg.get("bias"):                                # WN2D(bias=True) (waveflow.py:77): table order as the 1-D WN's, behind end.weight
            specs += [(p + "V.bias", (16 * Cd,), "bias"), (p + "start.bias", (C,), "bias")]
            for i in range(8):
                specs += [(p + "layers.%d.W.bias" % i, (2 * Cd,), "bias"), (p + "layers.%d.W_o.bias" % i, (Cs if i == 7 else C + Cs,), "bias")]
            specs.append((p + "end.bias", (2,), "bias"))
    if cfg.get("use_conv1x1"):                             # registered after WNs (waveflow.py:176-181): invconv1x1.{k}.weight [H, H, 1]
        specs += [("invconv1x1.%d.weight" % k, (H, H, 1), "orth") for k in range(cfg["flows"])]
    return specs


WF_CONFIGS = {
    # 8 rows (all height dilations 1), upsampling stride 32
    "wf8": dict(flows=3, n_group=8, n_mels=12, dilation_channels=32, residual_channels=32, skip_channels=32),
    # 64 rows: the shipped height-dilation pattern 1,2,4,8,16,1,2,4 (waveflow.py:85), stride 4, 3x3 taps reaching 32 rows up
    "wf64": dict(flows=2, n_group=64, n_mels=10, dilation_channels=32, residual_channels=32, skip_channels=64),
}
# use_conv1x1=True: an invertible 1x1 conv over the height axis replaces the flip between flows (waveflow.py:203-206)
WF_CONFIGS["wf8c"] = dict(WF_CONFIGS["wf8"], use_conv1x1=True)
WF_CONFIGS["wf64c"] = dict(WF_CONFIGS["wf64"], use_conv1x1=True)
# WN2D(bias=True) (no shipped config sets it): a bias on every conv of every WN2D
WF_CONFIGS["wf8b"] = dict(WF_CONFIGS["wf8"], bias=True)
WF_CONFIGS["wf64b"] = dict(WF_CONFIGS["wf64"], bias=True)
WF_SHAPES = {"wf8": (2, 8 * 96, 3), "wf64": (2, 64 * 24, 6)}       # (batch, samples, mel frames)
WF_SHAPES["wf8c"], WF_SHAPES["wf64c"] = WF_SHAPES["wf8"], WF_SHAPES["wf64"]
WF_SHAPES["wf8b"], WF_SHAPES["wf64b"] = WF_SHAPES["wf8"], WF_SHAPES["wf64"]


def waveflow_inputs(tag, B, N, F, n_mels):
    return uniform(tag + "/audio", (B, N), -1.0, 1.0), normal(tag + "/mel", (B, n_mels, F))


# STFTDecimate cases: (batch, samples, ratio) -- the shipped 2x / 3x models at their training segment, and a short ragged length
DECIMATE_CASES = {"r2_8192": (2, 8192, 2), "r3_8193": (1, 8193, 3), "r2_1500": (3, 1500, 2)}
