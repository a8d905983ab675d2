"""ctypes binding of csrc/libwgflow.so (the C ABI declared in include/wgflow.h).

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device the call
raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` (or `python build.py`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WGFLOW_LIB") or os.path.join(_HERE, "csrc", "libwgflow.so")   # WGFLOW_LIB: developer A/B builds
_LIB = None

ABI_VERSION = 9          # include/wgflow.h WG_ABI_VERSION (9: wg_reload_env, wg_stat_gate_part_launches)
ABI_SYMBOLS = [
    "wg_strerror", "wg_abi_version", "wg_param_count", "wg_packed_bytes", "wg_workspace_bytes",
    "wg_wn_param_count", "wg_wn_packed_bytes", "wg_coupling_workspace_bytes", "wg_invconv_workspace_bytes",
    "wg_workspace_init", "wg_pack_weights", "wg_wn_pack_weights", "wg_forward", "wg_inverse", "wg_backward",
    "wg_nll_loss", "wg_nll_loss_backward", "wg_invconv_apply", "wg_invconv_backward", "wg_coupling_apply",
    "wg_coupling_backward", "wg_upsample", "wg_wn_apply", "wg_wsr_cond", "wg_wsr_cond_backward", "wg_adam_step",
    "wg_wf_param_count", "wg_wf_packed_bytes", "wg_wf_workspace_bytes", "wg_wf_tape_bytes", "wg_wf_pack_weights", "wg_wf_upsample", "wg_wf_forward",
    "wg_wf_inverse", "wg_wf_backward", "wg_melspec_frames", "wg_melspec", "wg_lowpass_workspace_bytes", "wg_lowpass", "wg_train_step",
    "wg_nll_scratch_floats", "wg_train_scratch_floats",
    "wg_timer_create", "wg_timer_attach", "wg_timer_count", "wg_timer_read", "wg_timer_read_info", "wg_timer_destroy", "wg_stat_wgrad16t_launches",
    "wg_stat_layer_launches", "wg_layer_workspace_bytes", "wg_layer_apply", "wg_wf_wn_apply",
    "wg_timer_read_name", "wg_box_probe_bytes", "wg_box_probe", "wg_stat_layerg_launches", "wg_stat_gate_split_launches",
    "wg_wf_wn_backward", "wg_layer_backward_workspace_bytes", "wg_layer_backward", "wg_affine_apply", "wg_affine_backward",
    "wg_reload_env", "wg_stat_gate_part_launches", "wg_wsr_cond_pre",
]
K_CONV_STORE, K_CONV_GATE, K_CONV_RESSKIP, K_CONV_DGATE, K_WGRAD, K_LAYER, K_THIN = range(7)


class WgConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_flows", "n_group", "n_early_every", "n_early_size", "n_mels",
        "up_stride", "up_kernel", "up_pad", "res_ch", "dil_ch", "skip_ch", "depth", "radix", "precision", "reverse_mode", "keep_activations",
        "bias")]


class WgWfConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("flows", "n_group", "n_mels", "res_ch", "dil_ch", "skip_ch", "precision", "use_conv1x1", "bias")]


class WgWnDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_ch", "aux_ch", "res_ch", "dil_ch", "skip_ch", "depth", "radix", "precision", "bias")]


class WgLayerDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("res_ch", "dil_ch", "skip_ch", "radix", "dilation", "last_layer", "h_dilation", "rows")]


PREC_F32, PREC_BF16X3, PREC_BF16X3_PLANES = 0, 1, 2


def default_precision():
    """WG_PRECISION=f32|bf16x3|bf16x3p selects the arithmetic of the MFMA contractions (include/wgflow.h, WG_PREC_*)."""
    v = os.environ.get("WG_PRECISION", "bf16x3p").lower()
    if v in ("f32", "fp32", "0"):
        return PREC_F32
    if v in ("bf16x3", "1"):
        return PREC_BF16X3
    if v in ("bf16x3p", "bf16x3_planes", "2"):
        return PREC_BF16X3_PLANES
    raise WgError("WG_PRECISION must be f32 or bf16x3 (got %r)" % v)


class WgError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise WgError("HIP extension %s is missing -- build it first (python build.py); "
                      "this package has no CPU or eager fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, i, f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    cfgp, wnp = C.POINTER(WgConfig), C.POINTER(WgWnDims)
    L.wg_strerror.restype = C.c_char_p
    L.wg_strerror.argtypes = [i]
    L.wg_abi_version.restype = i
    if L.wg_abi_version() != ABI_VERSION:
        raise WgError("%s implements ABI revision %d, this package binds revision %d (include/wgflow.h WG_ABI_VERSION): rebuild it"
                      % (LIB_PATH, L.wg_abi_version(), ABI_VERSION))
    L.wg_param_count.argtypes = [cfgp]
    L.wg_packed_bytes.restype = sz
    L.wg_packed_bytes.argtypes = [cfgp]
    L.wg_workspace_bytes.restype = sz
    L.wg_workspace_bytes.argtypes = [cfgp, i, i, i]
    L.wg_wn_param_count.argtypes = [wnp]
    L.wg_wn_packed_bytes.restype = sz
    L.wg_wn_packed_bytes.argtypes = [wnp]
    L.wg_coupling_workspace_bytes.restype = sz
    L.wg_coupling_workspace_bytes.argtypes = [wnp, i, i, i]
    L.wg_invconv_workspace_bytes.restype = sz
    L.wg_invconv_workspace_bytes.argtypes = [i, i, i]
    L.wg_workspace_init.argtypes = [vp, sz, vp]
    L.wg_pack_weights.argtypes = [cfgp, vp, vp, vp]
    L.wg_wn_pack_weights.argtypes = [wnp, vp, vp, vp]
    L.wg_forward.argtypes = [cfgp, vp, vp, vp, i, i, i, vp, vp, vp, sz, vp]
    L.wg_inverse.argtypes = [cfgp, vp, vp, vp, i, i, i, vp, vp, vp, sz, vp]
    L.wg_backward.argtypes = [cfgp, vp, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, sz, vp, vp]
    L.wg_nll_scratch_floats.restype = sz
    L.wg_nll_scratch_floats.argtypes = [i]
    L.wg_train_scratch_floats.restype = sz
    L.wg_train_scratch_floats.argtypes = [i, i]
    L.wg_nll_loss.argtypes = [vp, vp, i, i, f, i, vp, vp, vp, vp]
    L.wg_nll_loss_backward.argtypes = [vp, i, i, f, i, vp, vp, vp, vp]
    L.wg_invconv_apply.argtypes = [vp, i, vp, i, i, i, vp, vp, vp, sz, vp]
    L.wg_invconv_backward.argtypes = [vp, i, vp, vp, vp, i, i, i, vp, vp, vp, vp, sz, vp]
    L.wg_coupling_apply.argtypes = [wnp, vp, vp, vp, i, i, i, vp, vp, vp, sz, vp]
    L.wg_coupling_backward.argtypes = [wnp, vp, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, sz, vp]
    L.wg_upsample.argtypes = [cfgp, vp, vp, i, i, i, vp, vp]
    L.wg_wn_apply.argtypes = [wnp, vp, vp, vp, i, i, vp, vp, vp, sz, vp]
    L.wg_wsr_cond.argtypes = [vp, i, i, vp, vp, vp, vp]
    L.wg_wsr_cond_backward.argtypes = [vp, i, i, vp, vp, vp, vp]
    L.wg_wsr_cond_pre.argtypes = [vp, i, i, vp, vp, vp]
    L.wg_adam_step.argtypes = [vp, vp, vp, vp, sz, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i, vp]
    wfp = C.POINTER(WgWfConfig)
    L.wg_wf_param_count.argtypes = [wfp]
    L.wg_wf_packed_bytes.restype = sz
    L.wg_wf_packed_bytes.argtypes = [wfp]
    L.wg_wf_workspace_bytes.restype = sz
    L.wg_wf_workspace_bytes.argtypes = [wfp, i, i, i]
    L.wg_wf_tape_bytes.restype = sz
    L.wg_wf_tape_bytes.argtypes = [wfp, i, i]
    L.wg_wf_pack_weights.argtypes = [wfp, vp, vp, vp]
    L.wg_wf_upsample.argtypes = [wfp, vp, vp, vp, i, i, i, vp, vp]
    L.wg_wf_forward.argtypes = [wfp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, sz, vp]
    L.wg_wf_inverse.argtypes = [wfp, vp, vp, vp, vp, i, i, i, vp, vp, vp, sz, vp]
    L.wg_wf_wn_apply.argtypes = [wfp, vp, vp, vp, vp, i, i, i, vp, vp, vp, sz, vp]
    L.wg_wf_wn_backward.argtypes = [wfp, vp, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, sz, vp]
    L.wg_wf_backward.argtypes = [wfp, vp, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, sz, vp]
    L.wg_melspec_frames.argtypes = [i, i, i]
    L.wg_melspec.argtypes = [vp, i, i, i, i, i, C.c_double, C.c_double, i, vp, vp, vp]
    L.wg_lowpass_workspace_bytes.restype = sz
    L.wg_lowpass_workspace_bytes.argtypes = [i, i, i, i]
    L.wg_lowpass.argtypes = [vp, i, i, i, i, i, i, vp, vp, sz, vp]
    L.wg_train_step.argtypes = [cfgp, vp, vp, vp, vp, i, i, i, f, i, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp]
    L.wg_timer_create.restype = vp
    L.wg_timer_create.argtypes = [i, i]
    L.wg_timer_attach.argtypes = [vp]
    L.wg_timer_attach.restype = None
    L.wg_timer_count.argtypes = [vp]
    L.wg_timer_read.argtypes = [vp, vp, i]
    L.wg_timer_read_info.argtypes = [vp, vp, i]
    L.wg_stat_wgrad16t_launches.restype = C.c_longlong
    L.wg_stat_wgrad16t_launches.argtypes = []
    L.wg_stat_layer_launches.restype = C.c_longlong
    L.wg_stat_layer_launches.argtypes = []
    L.wg_layer_workspace_bytes.restype = sz
    L.wg_layer_workspace_bytes.argtypes = [C.POINTER(WgLayerDims), i, i]
    L.wg_layer_apply.argtypes = [C.POINTER(WgLayerDims), vp, vp, vp, i, i, vp, vp, vp, sz, vp]
    L.wg_layer_backward_workspace_bytes.restype = C.c_size_t
    L.wg_layer_backward_workspace_bytes.argtypes = [C.POINTER(WgLayerDims), i, i]
    L.wg_layer_backward.argtypes = [C.POINTER(WgLayerDims), vp, vp, vp, vp, vp, i, i, vp, vp, vp, vp, sz, vp]
    L.wg_affine_apply.argtypes = [vp, vp, vp, sz, i, vp, vp]
    L.wg_affine_backward.argtypes = [vp, vp, vp, vp, vp, sz, i, vp, vp, vp, vp, vp]
    L.wg_timer_destroy.argtypes = [vp]
    L.wg_timer_destroy.restype = None
    L.wg_timer_read_name.argtypes = [vp, i, C.c_char_p, i]
    L.wg_box_probe_bytes.restype = C.c_size_t
    L.wg_box_probe_bytes.argtypes = []
    L.wg_box_probe.argtypes = [vp, i, vp, vp]
    L.wg_stat_layerg_launches.restype = C.c_longlong
    L.wg_stat_gate_split_launches.restype = C.c_longlong
    L.wg_stat_gate_part_launches.restype = C.c_longlong
    L.wg_reload_env.restype = None
    L.wg_reload_env.argtypes = []
    _LIB = L
    return L


def check(rc, what):
    if rc != 0:
        raise WgError("%s failed: %s (code %d)" % (what, lib().wg_strerror(rc).decode(), rc))
