"""ctypes/numpy front end of oracle/wf_oracle.c -- the WaveFlow oracle (TEST INFRASTRUCTURE, see the C file's header).

Parameter table = list of float32 arrays in the order of the reference model's named_parameters() for
WaveFlow(bias=False): upsampler.1.{bias, weight_g, weight_v}, then per flow
WNs.k.{V.weight_g, V.weight_v, start.weight_g, start.weight_v, layers.i.{W.weight_g, W.weight_v, W_o.weight_g, W_o.weight_v} x 8, end.weight},
then (use_conv1x1=True) invconv1x1.k.weight [H, H, 1] per flow.
"""
import ctypes as C
import os

import numpy as np

from . import wg_oracle as _wg

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("flows", "n_group", "n_mels", "res_ch", "dil_ch", "skip_ch", "use_conv1x1", "bias")]


def make_config(flows, n_group, n_mels, dilation_channels=256, residual_channels=256, skip_channels=256, use_conv1x1=False, bias=False, **_unused):
    """keyword names of the reference's WaveFlow(**arch.args) (model/waveflow.py:156-162, configs/waveflow_LJ_speech.json)"""
    return Config(flows, n_group, n_mels, residual_channels, dilation_channels, skip_channels, int(bool(use_conv1x1)), int(bool(bias)))


def _lib(double=False):
    key = "64" if double else "32"
    if key not in _LIBS:
        path = os.path.join(_HERE, "libwforacle64.so" if double else "libwforacle.so")
        if not os.path.exists(path):
            _wg.build()
        lib = C.CDLL(path)
        assert lib.wfo_real_bytes() == (8 if double else 4)
        _LIBS[key] = lib
    return _LIBS[key]


_f32, _ptr, _table, _check = _wg._f32, _wg._ptr, _wg._table, _wg._check


def param_count(cfg):
    return _lib().wfo_param_count(C.byref(cfg))


def forward(cfg, params, audio, mel, double=False):
    audio, mel = _f32(audio), _f32(mel)
    B, N = audio.shape
    tab, _keep = _table(params)
    z = np.empty((B, N), np.float32)
    logdet = np.empty((B,), np.float32)
    _check(_lib(double).wfo_forward(C.byref(cfg), tab, _ptr(audio), _ptr(mel), B, N, mel.shape[2], _ptr(z), _ptr(logdet)), "wf forward")
    return z, logdet


def inverse(cfg, params, z, mel, double=False):
    z, mel = _f32(z), _f32(mel)
    B, N = z.shape
    tab, _keep = _table(params)
    x = np.empty((B, N), np.float32)
    logdet = np.empty((B,), np.float32)
    _check(_lib(double).wfo_inverse(C.byref(cfg), tab, _ptr(z), _ptr(mel), B, N, mel.shape[2], _ptr(x), _ptr(logdet)), "wf inverse")
    return x, logdet


def train_step(cfg, params, audio, mel, sigma, need_dmel=False, double=False):
    """forward + NLL + backward.  Returns dict(z, logdet, loss, grads[list like params], dmel)."""
    audio, mel = _f32(audio), _f32(mel)
    B, N = audio.shape
    tab, _keep = _table(params)
    grads = [np.zeros_like(_f32(p)) for p in params]
    gtab = (C.POINTER(C.c_float) * len(grads))(*[_ptr(g) for g in grads])
    z = np.empty((B, N), np.float32)
    logdet = np.empty((B,), np.float32)
    lossv = C.c_float()
    dmel = np.empty_like(mel) if need_dmel else None
    _check(_lib(double).wfo_train_step(C.byref(cfg), tab, _ptr(audio), _ptr(mel), B, N, mel.shape[2], C.c_float(sigma),
                                       _ptr(z), _ptr(logdet), C.byref(lossv), gtab, _ptr(dmel)), "wf train_step")
    return dict(z=z, logdet=logdet, loss=float(lossv.value), grads=grads, dmel=dmel)
