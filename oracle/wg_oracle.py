"""ctypes/numpy front end of oracle/wg_oracle.c  (TEST INFRASTRUCTURE -- see the C file's header).

All arrays are float32 numpy, C-contiguous.  The parameter table is a list of arrays in the order of
the reference model's named_parameters() (see wg_oracle.c "model level"); `None` in a weight_g slot
means the conv carries a plain weight (after remove_weight_norm).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_flows", "n_group", "n_early_every", "n_early_size", "n_mels",
        "up_stride", "up_kernel", "up_pad",
        "res_ch", "dil_ch", "skip_ch", "depth", "radix", "bias")]


def make_config(flows, n_group, n_early_every, n_early_size, hop_size, n_mels,
                dilation_channels=256, residual_channels=256, skip_channels=256,
                depth=8, radix=3, bias=False, **_unused):
    """Same keyword names as the reference's WaveGlow(**arch.args) (model/waveglow.py:109-118)."""
    up = hop_size // n_group                      # waveglow.py:125
    k = up * 2 + 1                                # :126
    pad = k // 2 - up // 2                        # :128-129
    return Config(flows, n_group, n_early_every, n_early_size, n_mels, up, k, pad,
                  residual_channels, dilation_channels, skip_channels, depth, radix, int(bool(bias)))


def build(force=False):
    """Compile the oracle with the Makefile next to this file (gcc only)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, f)) for f in ("libwgoracle.so", "libwgoracle64.so", "libwforacle.so", "libwforacle64.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def _lib(double=False):
    key = "64" if double else "32"
    if key not in _LIBS:
        path = os.path.join(_HERE, "libwgoracle64.so" if double else "libwgoracle.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        assert lib.wgo_real_bytes() == (8 if double else 4)
        _LIBS[key] = lib
    return _LIBS[key]


def set_threads(n, double=False):
    """Caps the oracle's OpenMP team; returns the team size in effect."""
    return int(_lib(double).wgo_set_threads(int(n)))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else C.POINTER(C.c_float)()


def _table(arrs):
    keep = [None if a is None else _f32(a) for a in arrs]
    tab = (C.POINTER(C.c_float) * len(keep))(*[_ptr(a) for a in keep])
    return tab, keep


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed with code %d" % (what, rc))


def param_count(cfg):
    return _lib().wgo_param_count(C.byref(cfg))


def forward(cfg, params, audio, h, double=False, reverse_mode=False):
    """model.forward(x, h); reverse_mode=True is the architecture WaveGlow(reverse_mode=True) builds."""
    audio, h = _f32(audio), _f32(h)
    B, N = audio.shape
    F = h.shape[2]
    tab, _keep = _table(params)
    z = np.empty((B, N), np.float32)
    logdet = np.empty((B,), np.float32)
    fn = _lib(double).wgo_forward_rm if reverse_mode else _lib(double).wgo_forward
    _check(fn(C.byref(cfg), tab, _ptr(audio), _ptr(h), B, N, F, _ptr(z), _ptr(logdet)), "forward")
    return z, logdet


def inverse(cfg, params, z, h, double=False, reverse_mode=False):
    """model.reverse(z, h)"""
    z, h = _f32(z), _f32(h)
    B, N = z.shape
    F = h.shape[2]
    tab, _keep = _table(params)
    x = np.empty((B, N), np.float32)
    logdet = np.empty((B,), np.float32)
    fn = _lib(double).wgo_inverse_rm if reverse_mode else _lib(double).wgo_inverse
    _check(fn(C.byref(cfg), tab, _ptr(z), _ptr(h), B, N, F, _ptr(x), _ptr(logdet)), "inverse")
    return x, logdet


def loss(z, logdet, sigma, double=False):
    z, logdet = _f32(z), _f32(logdet)
    out = C.c_float()
    _check(_lib(double).wgo_loss(_ptr(z), _ptr(logdet), z.shape[0], z.shape[1], C.c_float(sigma), C.byref(out)), "loss")
    return float(out.value)


def train_step(cfg, params, audio, h, sigma, need_dh=False, double=False, reverse_mode=False):
    """forward + NLL + backward.  Returns dict(z, logdet, loss, grads[list like params], dh)."""
    audio, h = _f32(audio), _f32(h)
    B, N = audio.shape
    F = h.shape[2]
    tab, _keep = _table(params)
    grads = [None if p is None else np.zeros_like(_f32(p)) for p in params]
    gtab = (C.POINTER(C.c_float) * len(grads))(*[_ptr(g) for g in grads])
    z = np.empty((B, N), np.float32)
    logdet = np.empty((B,), np.float32)
    lossv = C.c_float()
    dh = np.empty_like(h) if need_dh else None
    fn = _lib(double).wgo_train_step_rm if reverse_mode else _lib(double).wgo_train_step
    _check(fn(C.byref(cfg), tab, _ptr(audio), _ptr(h), B, N, F, C.c_float(sigma),
                                       _ptr(z), _ptr(logdet), C.byref(lossv), gtab, _ptr(dh)), "train_step")
    return dict(z=z, logdet=logdet, loss=float(lossv.value), grads=grads, dh=dh)


def upsample(cfg, bias, g, v, h, T, double=False):
    h = _f32(h)
    B, _, F = h.shape
    bias, v = _f32(bias), _f32(v)
    g = None if g is None else _f32(g)
    y = np.empty((B, cfg.n_mels, T), np.float32)
    _check(_lib(double).wgo_upsample(C.byref(cfg), _ptr(bias), _ptr(g), _ptr(v), _ptr(h), B, F, T, _ptr(y)), "upsample")
    return y


# ---- block level -------------------------------------------------------------------------------

def invconv_forward(W, x, double=False):
    W, x = _f32(W).reshape(W.shape[0], W.shape[1]), _f32(x)
    B, c, T = x.shape
    z = np.empty_like(x)
    ld = C.c_float()
    _check(_lib(double).wgo_invconv_forward(_ptr(W), c, _ptr(x), B, T, _ptr(z), C.byref(ld)), "invconv_forward")
    return z, np.float32(ld.value)


def invconv_reverse(W, z, double=False):
    W, z = _f32(W).reshape(W.shape[0], W.shape[1]), _f32(z)
    B, c, T = z.shape
    x = np.empty_like(z)
    ld = C.c_float()
    _check(_lib(double).wgo_invconv_reverse(_ptr(W), c, _ptr(z), B, T, _ptr(x), C.byref(ld)), "invconv_reverse")
    return x, np.float32(ld.value)


def invconv_backward(W, z, dz, dlogdet, reverse=False, double=False):
    """Backward of Conv1x1Func (reverse=False) / InvConv1x1Func (reverse=True): from the block OUTPUT z and
    its gradient, returns (rebuilt input, d input, dW)."""
    W, z, dz = _f32(W).reshape(W.shape[0], W.shape[1]), _f32(z), _f32(dz)
    B, c, T = z.shape
    x, dx, dW = np.empty_like(z), np.empty_like(z), np.empty((c, c), np.float32)
    fn = _lib(double).wgo_invconv_reverse_backward if reverse else _lib(double).wgo_invconv_backward
    _check(fn(_ptr(W), c, _ptr(z), _ptr(dz), C.c_float(dlogdet), B, T, _ptr(x), _ptr(dx), _ptr(dW)), "invconv_backward")
    return x, dx, dW


def _wn_args(wn):
    return (wn["in_channels"], wn["aux_channels"], wn.get("residual_channels", 256), wn.get("dilation_channels", 256),
            wn.get("skip_channels", 256), wn.get("depth", 8), wn.get("radix", 3))


def coupling_apply(wn, params, x, y, reverse=False, double=False):
    """AffineCouplingBlock.forward_computation / reverse_computation; wn = WN ctor kwargs; params = 4+4*depth+1 arrays."""
    x, y = _f32(x), _f32(y)
    B, c, T = x.shape
    tab, _keep = _table(params)
    z = np.empty_like(x)
    ls = np.empty((B, c // 2, T), np.float32)
    _check(_lib(double).wgo_coupling_apply(*_wn_args(wn), tab, _ptr(x), _ptr(y), B, T, int(reverse), _ptr(z), _ptr(ls)), "coupling_apply")
    return z, ls


def coupling_backward(wn, params, z, y, dz, dlog_s, need_dy=True, reverse=False, double=False):
    """Backward of AffineCouplingFunc (reverse=False) / InvAffineCouplingFunc (reverse=True) from the block output."""
    z, y, dz, dlog_s = _f32(z), _f32(y), _f32(dz), _f32(dlog_s)
    B, c, T = z.shape
    tab, _keep = _table(params)
    grads = [None if p is None else np.zeros_like(_f32(p)) for p in params]
    gtab = (C.POINTER(C.c_float) * len(grads))(*[_ptr(g) for g in grads])
    x, dx = np.empty_like(z), np.empty_like(z)
    dy = np.empty_like(y) if need_dy else None
    fn = _lib(double).wgo_coupling_reverse_backward if reverse else _lib(double).wgo_coupling_backward
    _check(fn(*_wn_args(wn), tab, _ptr(z), _ptr(y), _ptr(dz), _ptr(dlog_s), B, T, _ptr(x), _ptr(dx), _ptr(dy), gtab), "coupling_backward")
    return dict(x=x, dx=dx, dy=dy, grads=grads)


# ---- WSRGlow conditioning front-end (model/wsrglow.py:37-50) -------------------------------------

WSR_COND = 8 * 400 + 9 * 51


def wsr_cond(c, mu_w, ang_w, double=False, return_idx=False):
    """cond[B,3659,L/8] = cat(mu-law embedding, |STFT16|, phase embedding).  c is NOT modified (the reference clips it in place)."""
    c, mu_w, ang_w = _f32(c), _f32(mu_w), _f32(ang_w)
    B, L = c.shape
    assert mu_w.shape == (256, 400) and ang_w.shape == (120, 50)
    F = L // 8
    cond = np.empty((B, WSR_COND, F), np.float32)
    mi = np.empty((B, L), np.int32)
    ai = np.empty((B, 9, F), np.int32)
    ip = C.POINTER(C.c_int32)
    _check(_lib(double).wgo_wsr_cond(_ptr(c), B, L, _ptr(mu_w), _ptr(ang_w), _ptr(cond),
                                     mi.ctypes.data_as(ip), ai.ctypes.data_as(ip)), "wsr_cond")
    return (cond, mi, ai) if return_idx else cond


def wsr_cond_backward(c, dcond, double=False):
    """-> (d mu_enc.1.weight [256,400], d angle_embed.embed.weight [120,50])."""
    c, dcond = _f32(c), _f32(dcond)
    B, L = c.shape
    assert dcond.shape == (B, WSR_COND, L // 8)
    dmu = np.empty((256, 400), np.float32)
    dang = np.empty((120, 50), np.float32)
    _check(_lib(double).wgo_wsr_cond_backward(_ptr(c), B, L, _ptr(dcond), _ptr(dmu), _ptr(dang)), "wsr_cond_backward")
    return dmu, dang
