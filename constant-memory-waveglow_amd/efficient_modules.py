"""Reversible blocks with the constant-memory protocol of the reference's model/efficient_modules.py,
executed by the HIP engine (include/wgflow.h: wg_invconv_*, wg_coupling_*).

Protocol (efficient_modules.py:33-35,71-75,113,135-136,225,236-237): with memory_efficient=True a block keeps
only (input handle, y, output); the caller-visible input storage is freed right after the forward pass and is
re-materialised in place during backward from the block's OUTPUT.  With memory_efficient=False the input is
simply left alone; the arithmetic (and the kernels) are the same.
"""
from typing import Tuple

import torch
from torch import Tensor, nn
from torch.autograd import Function

from . import engine
from ._lib import WgWnDims, WgError, default_precision as _default_precision
from .base import Reversible

__all__ = ["InvertibleConv1x1", "AffineCouplingBlock"]


def _free(t: Tensor):
    t.untyped_storage().resize_(0)


def _rematerialise(handle: Tensor, like: Tensor) -> Tensor:
    """Give `handle` (whose storage may have been freed) room for like.numel() floats and return the tensor the
    kernels should write the rebuilt input into."""
    need = like.numel() * like.element_size()
    if handle.untyped_storage().size() < need:
        handle.untyped_storage().resize_(need)
    return handle


class _Conv1x1(Function):
    """Conv1x1Func / InvConv1x1Func (efficient_modules.py:215-279) in one Function; `reverse` picks the pair."""

    @staticmethod
    def forward(ctx, x, weight, reverse):
        z, logdet = engine.invconv_apply(weight.detach(), x.detach(), reverse)
        ctx.reverse = reverse
        ctx.save_for_backward(x.data, weight, z)
        return z, logdet

    @staticmethod
    def backward(ctx, dz, dlogdet):
        x, weight, z = ctx.saved_tensors
        xo = _rematerialise(x, z)
        dx, dW = engine.invconv_backward(weight.detach(), z, dz, dlogdet, ctx.reverse, xo)
        return dx, dW.unsqueeze(-1), None


class InvertibleConv1x1(Reversible, nn.Conv1d):
    """Invertible 1x1 convolution z = W x with log|det| bookkeeping (efficient_modules.py:17-54)."""

    def __init__(self, c, memory_efficient=False, reverse_mode=False):
        super().__init__(in_channels=c, out_channels=c, kernel_size=1, bias=False, reverse_mode=reverse_mode)
        q = torch.linalg.qr(torch.randn(c, c))[0]          # orthogonal init, det forced positive (:22-26)
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        with torch.no_grad():
            self.weight.copy_(q.contiguous().unsqueeze(-1))
        self._memory_efficient = bool(memory_efficient)

    def _run(self, x: Tensor, reverse: bool) -> Tuple[Tensor, Tensor]:
        z, logdet = _Conv1x1.apply(x, self.weight, reverse)
        if self._memory_efficient:
            _free(x)
        return z, logdet

    def forward_computation(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        return self._run(x, False)

    def reverse_computation(self, z: Tensor) -> Tuple[Tensor, Tensor]:
        return self._run(z, True)


class _Coupling(Function):
    """AffineCouplingFunc / InvAffineCouplingFunc (efficient_modules.py:99-212)."""

    @staticmethod
    def forward(ctx, x, y, block, reverse, *weights):
        table = block.F.param_table()
        z, log_s = block._engine.apply([None if t is None else t.detach() for t in table], x.detach(), y.detach(), reverse)
        ctx.block, ctx.reverse = block, reverse
        ctx.save_for_backward(x.data, y, z)
        return z, log_s

    @staticmethod
    def backward(ctx, dz, dlog_s):
        x, y, z = ctx.saved_tensors
        block = ctx.block
        table = block.F.param_table()
        plist = list(block.F.parameters())
        need = [t is not None and t.requires_grad for t in table]
        xo = _rematerialise(x, z)
        dx, dy, grads = block._engine.backward([None if t is None else t.detach() for t in table], z, y, dz, dlog_s,
                                               ctx.reverse, need, ctx.needs_input_grad[1], xo)
        by_id = {id(t): g for t, g in zip(table, grads) if t is not None}
        return (dx, dy, None, None) + tuple(by_id.get(id(p)) for p in plist)


class AffineCouplingBlock(Reversible):
    """Affine coupling z_a = x_a, z_b = x_b * exp(log_s) + t with (log_s, t) = F(x_a, y)  (efficient_modules.py:57-96).
    `transform_type` must be this package's WN (the fused HIP transform net)."""

    def __init__(self, transform_type, memory_efficient=True, reverse_mode=False, **kwargs):
        super().__init__(reverse_mode)
        self.F = transform_type(**kwargs)
        if not hasattr(self.F, "hip_dims"):
            raise WgError("AffineCouplingBlock runs on the fused HIP WN kernels; transform_type must be "
                          "constant_memory_waveglow_amd.WN (got %r)" % (transform_type,))
        self._memory_efficient = bool(memory_efficient)
        self._engine = engine.CouplingEngine(WgWnDims(*self.F.hip_dims(), _default_precision(), int(getattr(self.F, "has_bias", False))))

    def _run(self, x: Tensor, y: Tensor, reverse: bool) -> Tuple[Tensor, Tensor]:
        z, log_s = _Coupling.apply(x, y, self, reverse, *self.F.parameters())
        if self._memory_efficient:
            _free(x)
        return z, log_s

    def forward_computation(self, x: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
        return self._run(x, y, False)

    def reverse_computation(self, z: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
        return self._run(z, y, True)
