"""WaveFlow on the HIP flow engine: same constructor, parameter names and forward/reverse/infer contract as the reference's
model/waveflow.py (use_conv1x1=False is the shipped configuration; True puts an InvertibleConv1x1(n_group) over the height axis
between the flows instead of the flip).  Audio [B, N] is viewed as [B, n_group (height), N/n_group (time)];
each flow runs WN2D (8 layers of 3x3 dilated convs, causal along the height axis) on rows 0..H-2 and transforms rows 1..H-1.

Module tree (state dicts interchange with the reference):
    upsampler.{0,1,2}     ReplicationPad1d((0,1)), ConvTranspose1d(n_mels, n_mels, 2s+1, s, padding s//2) + weight norm, LeakyReLU(0.4)
    WNs.{k}               WN2D: V, start, layers.{i}.{W, W_o}, end
    invconv1x1.{k}        InvertibleConv1x1(n_group)            (use_conv1x1=True only; registered after WNs as upstream)
"""
import warnings
from typing import Tuple

import torch
from torch import Tensor, nn
from torch.autograd import Function

from . import engine
from ._lib import WgError, WgWfConfig, default_precision
from .base import FlowBase
from .efficient_modules import InvertibleConv1x1
from .utils import add_weight_norms, conv_gv
from .waveglow import layer_bias_in, layer_bias_out


class NonCausalLayer2D(nn.Module):
    """One WN2D layer (waveflow.py:14-51): W = 3x3 conv with dilation (h_dilation, dilation), causal along the height axis, W_o = 1x1.  Inside
    WaveFlow its arithmetic runs in the WN kernels; called on its own, `forward` goes through wg_layer_apply (bias=True: the two biases are
    constants on y and on the outputs, added around that call)."""

    def __init__(self, h_dilation, dilation, dilation_channels, residual_channels, skip_channels, radix, bias, last_layer=False):
        super().__init__()
        self.h_pad_size = h_dilation * (radix - 1)
        self.pad_size = dilation * (radix - 1) // 2
        self.W = nn.Conv2d(residual_channels, dilation_channels * 2, kernel_size=radix, dilation=(h_dilation, dilation), bias=bias)
        self.chs_split = [skip_channels]
        if last_layer:
            self.W_o = nn.Conv2d(dilation_channels, skip_channels, 1, bias=bias)
        else:
            self.W_o = nn.Conv2d(dilation_channels, residual_channels + skip_channels, 1, bias=bias)
            self.chs_split.insert(0, residual_channels)

    def forward(self, x, y):
        """x [B, residual, H, W], y [B, 2 * dilation, 1, W] -> (x + res or None, skip) as waveflow.py:41-51 (wg_layer_apply: exact fp32 MFMA,
        forward only)."""
        from ._lib import WgLayerDims
        last = len(self.chs_split) == 1
        dims = WgLayerDims(self.W.in_channels, self.W.out_channels // 2, self.chs_split[-1], self.W.kernel_size[0], self.W.dilation[1], int(last),
                           self.W.dilation[0], x.shape[2])
        wg_, wv = conv_gv(self.W)
        og, ov = conv_gv(self.W_o)
        y = layer_bias_in(self.W, y.float())                                      # bias=True: as NonCausalLayer (waveglow.py)
        if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad or any(p.requires_grad for p in self.parameters())):
            out = engine.LayerFn.apply(x.float(), y, dims, wg_, wv, og, ov)           # differentiable like the module upstream (wg_layer_backward)
            return layer_bias_out(self.W_o, self.chs_split, (None, out[0]) if last else out)
        with torch.no_grad():
            return layer_bias_out(self.W_o, self.chs_split, engine.layer_apply(dims, [wg_, wv, og, ov], x.float(), y))


class WN2D(nn.Module):
    """Parameter container of WN2D (waveflow.py:70-135).  bias=True (no shipped config sets it): every conv carries a bias -- in the kernels one
    more K segment of ones behind the nine taps and the conditioning (csrc/wgflow.hip WnD::bias)."""

    H_DILATIONS = {8: [1] * 8, 16: [1] * 8, 32: [1, 2, 4] * 2 + [1, 2], 64: [1, 2, 4, 8, 16, 1, 2, 4], 128: [1, 2, 4, 8, 16, 32, 64, 1]}

    def __init__(self, n_group, aux_channels, dilation_channels=256, residual_channels=256, skip_channels=256, bias=False, zero_init=True):
        super().__init__()
        self.has_bias = bool(bias)
        self.h_dilations = self.H_DILATIONS[n_group]
        self.dilations = [2 ** i for i in range(8)]
        self.n_group = n_group
        self.res_chs, self.dil_chs, self.skp_chs, self.aux_chs = residual_channels, dilation_channels, skip_channels, aux_channels
        self.r_field = sum(self.dilations) * 2 + 1
        self.h_r_field = sum(self.h_dilations) * 2 + 1
        self.V = nn.Conv1d(aux_channels, dilation_channels * 2 * 8, 1, bias=bias)
        self.V.apply(add_weight_norms)
        self.start = nn.Conv2d(1, residual_channels, 1, bias=bias)
        self.start.apply(add_weight_norms)
        self.layers = nn.ModuleList(
            NonCausalLayer2D(hd, d, dilation_channels, residual_channels, skip_channels, 3, bias, last_layer=(i == 7))
            for i, (hd, d) in enumerate(zip(self.h_dilations, self.dilations)))
        self.layers.apply(add_weight_norms)
        self.end = nn.Conv2d(skip_channels, 2, 1, bias=bias)
        if zero_init:
            self.end.weight.data.zero_()
            if bias:
                self.end.bias.data.zero_()

    def param_table(self):
        """the flow's part of the C-ABI table (include/wgflow.h wg_wf_config): 37 weights, then -- bias=True -- the 19 biases"""
        tab = list(conv_gv(self.V)) + list(conv_gv(self.start))
        for layer in self.layers:
            tab += list(conv_gv(layer.W)) + list(conv_gv(layer.W_o))
        tab.append(self.end.weight)
        if self.has_bias:
            tab += [self.V.bias, self.start.bias]
            for layer in self.layers:
                tab += [layer.W.bias, layer.W_o.bias]
            tab.append(self.end.bias)
        return tab

    def forward(self, x, y):
        """x [B, 1, rows <= n_group, W], y [B, aux, W] -> (log_s, t), each [B, 1, rows, W] (waveflow.py:128-135).  Differentiable like
        the module upstream: with autograd on the call is a node whose backward is wg_wf_wn_backward (_WN2DFn); inside WaveFlow the
        gradients flow through wg_wf_backward."""
        if getattr(self, "_engine", None) is None:
            self._engine = engine.WN2DEngine(WgWfConfig(1, self.n_group, self.aux_chs, self.res_chs, self.dil_chs, self.skp_chs,
                                                         default_precision(), 0, int(self.has_bias)))
        if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _WN2DFn.apply(x.float(), y.float(), self, *self.parameters())
        with torch.no_grad():
            return self._engine.apply([None if p is None else p.detach() for p in self.param_table()], x.float(), y.float())


class _WN2DFn(Function):
    """WN2D.forward on its own as an autograd node: forward = wg_wf_wn_apply, backward = wg_wf_wn_backward (the recompute with the
    layers kept, then the WN backward seeded with the gradients of log_s and t)."""

    @staticmethod
    def forward(ctx, x, y, wn, *weights):
        table = wn.param_table()
        log_s, t = wn._engine.apply([None if p is None else p.detach() for p in table], x.detach(), y.detach())
        ctx.wn = wn
        ctx.save_for_backward(x, y)
        return log_s, t

    @staticmethod
    def backward(ctx, dlog_s, dt):
        x, y = ctx.saved_tensors
        wn = ctx.wn
        table = wn.param_table()
        need = [p is not None and p.requires_grad for p in table]
        dx, dy, grads = wn._engine.backward([None if p is None else p.detach() for p in table], x.detach(), y.detach(), dlog_s, dt, need,
                                            ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        by_id = {id(p): g for p, g in zip(table, grads) if p is not None}
        return (dx, dy, None) + tuple(by_id.get(id(p)) for p in wn.parameters())


class _WaveFlowFn(Function):
    """forward = wg_wf_forward (which tapes every flow's input), backward = wg_wf_backward (per-flow recompute from the tape)."""

    @staticmethod
    def forward(ctx, model, x, h, *params):
        table = [None if t is None else t.detach() for t in model.param_table()]
        # (Function.forward runs with grad mode off; whether a backward can follow is what needs_input_grad says)
        z, logdet, tape = model._engine.forward(table, x.detach(), h.detach(), keep_tape=any(ctx.needs_input_grad))
        ctx.model, ctx.tape = model, tape
        # (reverse_mode: the engine was handed M = W^-1 of the 1x1 weights AS THEY ARE NOW; the backward's chain rule must use the same M)
        ctx.inv_key = getattr(model, "_inv_key", None) if (model._reverse_mode and model._mix_modules()) else None
        ctx.save_for_backward(h)
        return z, logdet

    @staticmethod
    def backward(ctx, dz, dlogdet):
        (h,) = ctx.saved_tensors
        model = ctx.model
        table = model.param_table()
        if ctx.inv_key is not None and getattr(model, "_inv_key", None) != ctx.inv_key:
            # param_table() has just rewritten the cached W^-1 in place from CHANGED 1x1 weights: dW = -M^T G M^T with that M would be a
            # gradient of another function than the one the tape was produced with (the non-reverse path reads the live weights, where
            # autograd's version counters flag the same mistake)
            raise RuntimeError("WaveFlow(reverse_mode=True): a 1x1 weight was modified between forward and backward "
                               "(or another forward ran in between); run backward before changing the weights")
        grads, dmel, dx = model._engine.backward([None if t is None else t.detach() for t in table], ctx.tape, h, dz, dlogdet,
                                                 ctx.needs_input_grad[2], ctx.needs_input_grad[1])
        by_id = {id(t): g for t, g in zip(table, grads) if t is not None}
        mix = model._mix_modules()
        for m, t in zip(mix, table[len(table) - len(mix):]):
            if t is not m.weight:                               # reverse_mode: the engine was handed M = W^-1, so dL/dW = -M^T (dL/dM) M^T
                mt = t.squeeze(-1).double().t()
                by_id[id(m.weight)] = (-(mt @ by_id[id(t)].squeeze(-1).double() @ mt)).float().unsqueeze(-1)
        return (None, dx, dmel) + tuple(by_id.get(id(p)) for p in model.parameters())


class WaveFlow(FlowBase):
    def __init__(self, flows, n_group, n_mels, use_conv1x1, memory_efficient, reverse_mode=False, **kwargs):
        super().__init__(256, reverse_mode)
        # reverse_mode=True (base.py:20-28): `forward` is then the row-by-row loop (wg_wf_inverse: no autograd -- the reference trains that
        # direction through reverse_mode_forward's buffers, this engine does not) and `reverse` / `infer` the parallel map (wg_wf_forward,
        # differentiable); the 1x1 convs, when present, swap too: W in the row loop, W^-1 in the parallel map (efficient_modules.py:30-56).
        self.flows, self.n_group, self.n_mels = flows, n_group, n_mels
        self.sub_sr = self._hop_length // n_group
        self.upsampler = nn.Sequential(
            nn.ReplicationPad1d((0, 1)),
            nn.ConvTranspose1d(n_mels, n_mels, self.sub_sr * 2 + 1, self.sub_sr, padding=self.sub_sr // 2),
            nn.LeakyReLU(0.4, True))
        self.upsampler.apply(add_weight_norms)
        self.WNs = nn.ModuleList(WN2D(n_group, n_mels, **kwargs) for _ in range(flows))
        if use_conv1x1:                                          # waveflow.py:176-181 (the blocks are parameter containers here: the
            self.invconv1x1 = nn.ModuleList(                     # 1x1 over the height axis runs inside wg_wf_forward / _inverse / _backward)
                InvertibleConv1x1(n_group, memory_efficient=memory_efficient, reverse_mode=reverse_mode) for _ in range(flows))
        wn0 = self.WNs[0]
        self._engine = engine.WaveFlowEngine(WgWfConfig(flows, n_group, n_mels, wn0.res_chs, wn0.dil_chs, wn0.skp_chs, default_precision(),
                                                        int(bool(use_conv1x1)), int(wn0.has_bias)))

    def param_table(self):
        """C-ABI parameter table (include/wgflow.h): upsampler.1 bias, g, v; per flow the WN2D table."""
        up = self.upsampler[1]
        g, v = conv_gv(up)                       # (None, weight) after remove_weight_norms (inference.py:19-22): plain weights
        tab = [up.bias, g, v]
        for wn in self.WNs:
            tab += wn.param_table()
        mix = self._mix_modules()
        # the engine applies its 1x1 matrix in the parallel map and the inverse of it in the row loop: under reverse_mode hand it W^-1
        return tab + (self._inverse_mats(mix) if mix and self._reverse_mode else [m.weight for m in mix])

    def _mix_modules(self):
        return list(self.invconv1x1) if hasattr(self, "invconv1x1") else []

    def _inverse_mats(self, mix):
        """W^-1 of every 1x1, recomputed when a weight changed.  The tensors live as long as the model and are rewritten IN PLACE, so that
        their version counters tell the engine's pack cache (a fresh temporary could reuse a freed address at version 0)."""
        key = tuple((m.weight.data_ptr(), m.weight._version) for m in mix)
        if getattr(self, "_inv_key", None) != key:
            new = [torch.linalg.inv(m.weight.detach().squeeze(-1).double()).float().contiguous().unsqueeze(-1) for m in mix]
            old = getattr(self, "_inv", None)
            if old is not None and old[0].device == new[0].device:
                for a, b in zip(old, new):
                    a.copy_(b)
            else:
                self._inv = new
            self._inv_key = key
        return self._inv

    def _check(self, x: Tensor, h: Tensor):
        if x.dim() != 2 or h.dim() != 3:
            raise WgError("expected audio [B, N] and conditioning [B, n_mels, frames]")
        s = self.sub_sr
        assert x.size(1) // self.n_group <= h.size(2) * s - 2 * (s // 2) + 2 * s + 1

    def _upsample_h(self, h):
        """the conditioning at the flow's time resolution (waveflow.py:255-257); no autograd here (it is part of wg_wf_forward)"""
        return self._engine.upsample([None if t is None else t.detach() for t in self.param_table()], h.detach().float())

    def forward_computation(self, x: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        self._check(x, h)
        return _WaveFlowFn.apply(self, x, h, *self.parameters())

    def reverse_computation(self, z: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        self._check(z, h)
        if torch.is_grad_enabled() and (z.requires_grad or h.requires_grad or
                                        (self._reverse_mode and any(p.requires_grad for p in self.parameters()))):
            warnings.warn("WaveFlow's row-by-row direction runs without autograd in the HIP engine", stacklevel=3)
        return self._engine.inverse([None if t is None else t.detach() for t in self.param_table()], z.detach(), h.detach())
