"""WaveGlow on the HIP flow engine: same constructor, parameter names and forward/reverse/infer contract as the
reference's model/waveglow.py, but the computation is one call into libwgflow.so per direction.

Module tree (so state dicts interchange with the reference):
    upsampler                      ConvTranspose1d(n_mels, n_mels, 2u+1, stride u, groups n_mels) + weight norm
    invconv1x1.{k}                 InvertibleConv1x1(c_k)
    WNs.{k}.F                      WN(in=c_k/2, aux=n_mels, ...): V, start, layers.{i}.{W, W_o}, end
"""
import warnings
from typing import Tuple

import torch
from torch import Tensor, nn
from torch.autograd import Function

from . import engine
from ._lib import WgError
from .base import FlowBase
from .efficient_modules import AffineCouplingBlock, InvertibleConv1x1
from .utils import SlotTable, add_weight_norms, conv_gv, conv_gv_slots


def fused_gate(x1: Tensor, x2: Tensor) -> Tensor:
    """tanh(x1) * sigmoid(x2) -- the WN gate (waveglow.py:13-15).  Inside the engine this is the epilogue of the
    dilated-conv kernel; the function is kept for API parity and device tensors only."""
    engine.require_device(x1, x2)
    return torch.tanh(x1) * torch.sigmoid(x2)


def layer_bias_in(conv, y):
    """y + the bias of a stand-alone layer's W (one value per conditioning channel, broadcast over batch and positions)."""
    if conv.bias is None:
        return y
    return y + conv.bias.to(y.dtype).view(1, -1, *([1] * (y.dim() - 2)))


def layer_bias_out(conv, chs_split, out):
    """(res, skip) + the bias of a stand-alone layer's W_o, split like its output channels (waveglow.py:45)."""
    if conv.bias is None:
        return out
    res, skip = out
    parts = conv.bias.split(chs_split)
    shape = (1, -1) + (1,) * (skip.dim() - 2)
    skip = skip + parts[-1].to(skip.dtype).view(shape)
    if res is not None:
        res = res + parts[0].to(res.dtype).view(shape)
    return res, skip


class NonCausalLayer(nn.Module):
    """One WN layer: W = dilated conv (residual -> 2*dilation channels), W_o = 1x1 (dilation -> residual+skip, or skip only on the
    last layer)  (waveglow.py:18-46).  Inside a WN its arithmetic runs in the WN kernels; called on its own, `forward` runs the two
    products through the C ABI (wg_layer_apply: exact fp32 MFMA, any dilation, forward only -- the class has no backward upstream
    either: gradients flow through AffineCouplingBlock).  bias=True: W's bias is a constant on y, W_o's one on the outputs, added around
    that call (layer_bias_in / layer_bias_out)."""

    def __init__(self, dilation, dilation_channels, residual_channels, skip_channels, radix, bias, last_layer=False):
        super().__init__()
        self.W = nn.Conv1d(residual_channels, 2 * dilation_channels, kernel_size=radix, dilation=dilation,
                           padding=dilation * (radix - 1) // 2, bias=bias)
        out_ch = skip_channels if last_layer else residual_channels + skip_channels
        self.W_o = nn.Conv1d(dilation_channels, out_ch, 1, bias=bias)
        self.chs_split = [skip_channels] if last_layer else [residual_channels, skip_channels]

    def forward(self, x, y):
        """x [B, residual, T], y [B, 2 * dilation, T] (this layer's slice of the conditioning projection) -> (x + res or None, skip)
        as waveglow.py:41-46."""
        from ._lib import WgLayerDims
        last = len(self.chs_split) == 1
        dims = WgLayerDims(self.W.in_channels, self.W.out_channels // 2, self.chs_split[-1], self.W.kernel_size[0], self.W.dilation[0], int(last), 0, 0)
        wg_, wv = conv_gv(self.W)
        og, ov = conv_gv(self.W_o)
        # bias=True: xy = W(x) + y, so W's bias is a constant added to y's channels, and W_o's a constant on (res, skip); their gradients are
        # the sums autograd takes of d y and of the output gradients (inside WN the biases ride in the kernels' K segments instead)
        y = layer_bias_in(self.W, y)
        if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad or any(p.requires_grad for p in self.parameters())):
            out = engine.LayerFn.apply(x, y, dims, wg_, wv, og, ov)            # differentiable like the module upstream (wg_layer_backward)
            return layer_bias_out(self.W_o, self.chs_split, (None, out[0]) if last else out)
        with torch.no_grad():
            return layer_bias_out(self.W_o, self.chs_split, engine.layer_apply(dims, [wg_, wv, og, ov], x, y))


class WN(nn.Module):
    """Non-causal WaveNet-like transform net (waveglow.py:49-105): (log_s, t) = WN(x, y)."""

    def __init__(self, in_channels, aux_channels, dilation_channels=256, residual_channels=256, skip_channels=256,
                 depth=8, radix=3, bias=False, zero_init=True):
        super().__init__()
        self.has_bias = bool(bias)
        self.dilations = [2 ** i for i in range(depth)]
        self.in_chs, self.aux_chs = in_channels, aux_channels
        self.res_chs, self.dil_chs, self.skp_chs, self.rdx = residual_channels, dilation_channels, skip_channels, radix
        self.r_field = sum(self.dilations) + 1

        self.V = nn.Conv1d(aux_channels, 2 * dilation_channels * depth, 1, bias=bias)
        self.V.apply(add_weight_norms)
        self.start = nn.Conv1d(in_channels, residual_channels, 1, bias=bias)
        self.start.apply(add_weight_norms)
        self.layers = nn.ModuleList(
            NonCausalLayer(d, dilation_channels, residual_channels, skip_channels, radix, bias, last_layer=(i == depth - 1))
            for i, d in enumerate(self.dilations))
        self.layers.apply(add_weight_norms)
        self.end = nn.Conv1d(skip_channels, 2 * in_channels, 1, bias=bias)
        if zero_init:
            nn.init.zeros_(self.end.weight)
            if bias:
                nn.init.zeros_(self.end.bias)
        self._engine = None
        self._table = None

    def hip_dims(self):
        return (self.in_chs, self.aux_chs, self.res_chs, self.dil_chs, self.skp_chs, len(self.layers), self.rdx)

    def param_slots(self):
        tab = list(conv_gv_slots(self.V)) + list(conv_gv_slots(self.start))
        for layer in self.layers:
            tab += list(conv_gv_slots(layer.W)) + list(conv_gv_slots(layer.W_o))
        tab.append((self.end._parameters, "weight"))
        if self.has_bias:                                  # behind `end`: V, start, per layer (W, W_o), end  (wg_config.bias)
            tab += [(self.V._parameters, "bias"), (self.start._parameters, "bias")]
            for layer in self.layers:
                tab += [(layer.W._parameters, "bias"), (layer.W_o._parameters, "bias")]
            tab.append((self.end._parameters, "bias"))
        return tab

    def param_table(self):
        """C-ABI order: V(g,v) start(g,v) [W(g,v) W_o(g,v)]*depth end [+ the biases, WN(bias=True)]  (include/wgflow.h)."""
        if self._table is None:
            self._table = SlotTable("param_slots")
        return self._table(self)

    def forward(self, x, y):
        """(log_s, t) = WN(x, y) as waveglow.py:98-105 -- an ordinary differentiable module upstream, and here: called on its own with
        autograd on, the call is a node whose backward is the coupling block's (wg_coupling_backward seeded with the gradients of
        log_s and t, see _WNFn)."""
        if self._engine is None:
            from ._lib import WgWnDims, default_precision
            self._engine = engine.CouplingEngine(WgWnDims(*self.hip_dims(), default_precision(), int(self.has_bias)))
        if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _WNFn.apply(x, y, self, *self.parameters())
        return self._engine.wn([None if t is None else t.detach() for t in self.param_table()], x.detach(), y.detach())


class _WNFn(Function):
    """WN.forward on its own as an autograd node.  The backward is the coupling block's, unchanged: for z = cat(x, t) the block's
    `x_b = (z_b - t) / exp(log_s)` is exactly zero (the recompute gives the same t), so the seeds efficient_modules.py:143-144 forms,
    `d log_s = dz_b * x_b * exp(log_s) + dlog_s` and `d t = dz_b`, are the gradients handed in here when dz = cat(0, dt) and dlog_s is
    the gradient of log_s; rows [0, ic) of the block's dx are the gradient of x."""

    @staticmethod
    def forward(ctx, x, y, wn, *weights):
        table = wn.param_table()
        log_s, t = wn._engine.wn([None if p is None else p.detach() for p in table], x.detach(), y.detach())
        ctx.wn = wn
        ctx.save_for_backward(x, y, t)
        return log_s, t

    @staticmethod
    def backward(ctx, dlog_s, dt):
        x, y, t = ctx.saved_tensors
        wn = ctx.wn
        table = wn.param_table()
        need = [p is not None and p.requires_grad for p in table]
        z = torch.cat((x.detach(), t), 1)
        dz = torch.cat((torch.zeros_like(dt), dt), 1)
        dx, dy, grads = wn._engine.backward([None if p is None else p.detach() for p in table], z, y.detach(), dz, dlog_s, False, need,
                                            ctx.needs_input_grad[1], torch.empty_like(z))
        by_id = {id(p): g for p, g in zip(table, grads) if p is not None}
        return (dx[:, :x.size(1)] if ctx.needs_input_grad[0] else None, dy, None) + tuple(by_id.get(id(p)) for p in wn.parameters())


class _WaveGlowFn(Function):
    """Whole-model autograd node: forward = wg_forward; backward = wg_backward, which walks the flows last to first
    and rebuilds every block input from its output (constant activation memory in the number of flows).  With
    memory_efficient=False the WN activations of all flows are kept between the two calls instead (more memory, no
    recompute); the block inputs are still rebuilt from the outputs, which costs nothing extra."""

    @staticmethod
    def forward(ctx, model, x, h, *params):
        table = [None if t is None else t.detach() for t in model.param_table()]
        ctx.model, ctx.kept = model, None
        if model.mem_efficient:
            z, logdet = model._engine.run(table, x.detach(), h.detach(), False)
        else:
            # memory_efficient=False: every flow's WN layers stay in the engine's workspace and the backward reads them
            # instead of recomputing each WN (the reference's plain-autograd blocks, efficient_modules.py:33-35,71-75)
            z, logdet, ctx.kept = model._engine.run_keep(table, x.detach(), h.detach())
        ctx.save_for_backward(z, h)
        return z, logdet

    @staticmethod
    def backward(ctx, dz, dlogdet):
        z, h = ctx.saved_tensors
        model = ctx.model
        table = model.param_table()
        need = [t is not None and t.requires_grad for t in table]
        grads, dh, dx, _ = model._engine.backward([None if t is None else t.detach() for t in table], z, h, dz, dlogdet,
                                                  need, ctx.needs_input_grad[2], ctx.needs_input_grad[1], kept=ctx.kept)
        ctx.kept = None
        by_id = {id(t): g for t, g in zip(table, grads) if t is not None}
        return (None, dx, dh) + tuple(by_id.get(id(p)) for p in model.parameters())


class WaveGlow(FlowBase):
    def __init__(self, flows, n_group, n_early_every, n_early_size, hop_size, n_mels, memory_efficient,
                 reverse_mode=False, **kwargs):
        super().__init__(hop_size, reverse_mode)
        self.n_group, self.n_early_every, self.n_early_size = n_group, n_early_every, n_early_size
        self.n_mels, self.mem_efficient = n_mels, memory_efficient

        self.upsample_factor = hop_size // n_group
        k = 2 * self.upsample_factor + 1
        self.upsampler = nn.ConvTranspose1d(n_mels, n_mels, k, self.upsample_factor,
                                            padding=k // 2 - self.upsample_factor // 2, groups=n_mels)
        self.upsampler.apply(add_weight_norms)

        self.invconv1x1 = nn.ModuleList()
        self.WNs = nn.ModuleList()
        self.z_split_sizes = []
        c = n_group
        for k_flow in range(flows):
            if k_flow and k_flow % n_early_every == 0:       # early output: n_early_size channels leave the flow
                c -= n_early_size
                self.z_split_sizes.append(n_early_size)
            self.invconv1x1.append(InvertibleConv1x1(c, memory_efficient=memory_efficient, reverse_mode=reverse_mode))
            self.WNs.append(AffineCouplingBlock(WN, memory_efficient=memory_efficient, reverse_mode=reverse_mode,
                                                in_channels=c // 2, aux_channels=n_mels, **kwargs))
        self.z_split_sizes.append(c)

        self._half_table, self._half_key = None, None          # fp32 copies of half parameters (inference --half)
        self._table = SlotTable("param_slots")
        wn0 = self.WNs[0].F
        self._engine = engine.ModelEngine(engine.make_config(
            flows, n_group, n_early_every, n_early_size, hop_size, n_mels,
            wn0.res_chs, wn0.dil_chs, wn0.skp_chs, len(wn0.layers), wn0.rdx, reverse_mode=reverse_mode, bias=wn0.has_bias))

    def param_slots(self):
        g, v = conv_gv_slots(self.upsampler)
        tab = [(self.upsampler._parameters, "bias"), g, v] + [(m._parameters, "weight") for m in self.invconv1x1]
        for blk in self.WNs:
            tab += blk.F.param_slots()
        return tab

    def param_table(self):
        """C-ABI parameter table (include/wgflow.h): upsampler bias,g,v; 1x1 weights; per flow the WN table."""
        return self._table(self)

    def _check(self, x: Tensor, h: Tensor):
        if x.dim() != 2 or h.dim() != 3:
            raise WgError("expected audio [B, N] and conditioning [B, n_mels, frames]")
        T = x.size(1) // self.n_group
        up_len = (h.size(2) - 1) * self.upsample_factor - 2 * self.upsampler.padding[0] + self.upsampler.kernel_size[0]
        assert T <= up_len            # same contract as the reference's assert (waveglow.py:156,187)

    # The engine's wg_forward is "what model.forward computes" and wg_inverse "what model.reverse computes" for the
    # architecture selected by reverse_mode.  Reversible.forward dispatches to reverse_computation when reverse_mode is set
    # (base.py:20-28 upstream), so the differentiable direction lives in whichever method model.forward lands on.
    def _train_direction(self, x: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        self._check(x, h)
        return _WaveGlowFn.apply(self, x, h, *self.parameters())

    def _sample_direction(self, z: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        self._check(z, h)
        if torch.is_grad_enabled() and (z.requires_grad or h.requires_grad):
            warnings.warn("WaveGlow.reverse runs without autograd in the HIP engine", stacklevel=3)
        table = self.param_table()                 # (the engine reads addresses and versions only: no autograd through this call)
        if z.dtype == torch.float16 or h.dtype == torch.float16 or any(t is not None and t.dtype == torch.float16 for t in table):
            # `inference.py --half` (model.half(), cond.half(), inference.py:33-36): the engine computes in fp32, so half tensors are
            # widened on the way in and the result is narrowed on the way out -- the half-precision storage contract, fp32 arithmetic
            if self._half_table is None or self._half_key != tuple((t.data_ptr(), t._version) for t in table if t is not None):
                self._half_key = tuple((t.data_ptr(), t._version) for t in table if t is not None)
                self._half_table = [None if t is None else t.detach().float() for t in table]
            x, logdet = self._engine.run(self._half_table, z.detach().float(), h.detach().float(), True)
            return x.to(z.dtype), logdet.to(z.dtype)
        return self._engine.run(table, z.detach(), h.detach(), True)

    def forward_computation(self, x: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        return self._sample_direction(x, h) if self._reverse_mode else self._train_direction(x, h)

    def reverse_computation(self, z: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        return self._train_direction(z, h) if self._reverse_mode else self._sample_direction(z, h)

    def _upsample_h(self, h):
        T = (h.size(2) - 1) * self.upsample_factor - 2 * self.upsampler.padding[0] + self.upsampler.kernel_size[0]
        return self._engine.upsample([None if t is None else t.detach() for t in self.param_table()], h.detach(), T)
