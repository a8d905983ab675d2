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


class _GenericCoupling(Function):
    """AffineCouplingFunc / InvAffineCouplingFunc (efficient_modules.py:99-212) around ANY transform module F(x_a, y) -> (log_s, t): the
    transform runs as the caller's torch module (no graph kept in forward; recomputed with autograd on in backward, as upstream :127-130),
    the block's own arithmetic -- the affine map, the input rebuilt from the output, the seeds of the transform's backward -- runs in the
    library (wg_affine_apply / wg_affine_backward)."""

    @staticmethod
    def forward(ctx, x, y, block, reverse, *weights):
        ic = x.size(1) // 2
        xa, xb = x[:, :ic].contiguous(), x[:, ic:].contiguous()
        with torch.no_grad():
            log_s, t = block.F(xa, y)
            # (a transform under autocast may return half precision: the block's arithmetic is float32, as upstream's custom_fwd leaves it)
            log_s, t = log_s.float(), t.float()
            zb = engine.affine_apply(xb.float(), log_s, t, reverse)
            z = torch.cat((xa.float(), zb), 1)
        ctx.block, ctx.reverse = block, reverse
        ctx.save_for_backward(x.data, y, z)
        return z, (-log_s if reverse else log_s)

    @staticmethod
    def backward(ctx, dz, dlog_s):
        x, y, z = ctx.saved_tensors
        F, reverse = ctx.block.F, ctx.reverse
        ic = z.size(1) // 2
        za = z[:, :ic].detach().contiguous().requires_grad_(True)
        ya = y.detach().requires_grad_(True) if ctx.needs_input_grad[1] else y.detach()
        with torch.enable_grad():
            log_s, t = F(za, ya)                                                       # recompute :127-130 / :189-192
        rebuilt, g_ls, g_t, din = engine.affine_backward(z[:, ic:], log_s.detach().float(), t.detach().float(), dz[:, ic:].float(),
                                                         None if dlog_s is None else dlog_s.float(), reverse)
        with torch.no_grad():
            torch.cat((za.detach(), rebuilt), 1, out=_rematerialise(x, z))             # the freed input, in place (:135-136 / :197-198)
        plist = [p for p in F.parameters() if p.requires_grad]
        inputs = [za] + plist + ([ya] if ctx.needs_input_grad[1] else [])
        grads = torch.autograd.grad([log_s, t], inputs, [g_ls.to(log_s.dtype), g_t.to(t.dtype)], allow_unused=True)  # :140-144 / :200-204
        da = dz[:, :ic] + (grads[0] if grads[0] is not None else 0)
        dy = grads[-1] if ctx.needs_input_grad[1] else None
        by_id = {id(p): g for p, g in zip(plist, grads[1:1 + len(plist)])}
        return (torch.cat((da, din), 1), dy, None, None) + tuple(by_id.get(id(p)) for p in F.parameters())


class _AffineMap(Function):
    """The affine map of a coupling block on its own, for a block that keeps its graph (memory_efficient=False with a transform that is
    not this package's WN): (half, log_s, t) -> (half * exp(log_s) + t, log_s), or reversed ((half - t) / exp(log_s), -log_s)
    (efficient_modules.py:77-82 / :91-96).  Same kernels as _GenericCoupling (wg_affine_apply / wg_affine_backward); nothing is freed,
    nothing is written in place -- upstream runs this case under plain autograd and never touches the caller's x."""

    @staticmethod
    def forward(ctx, half, log_s, t, reverse):
        half, log_s, t = half.float(), log_s.float(), t.float()
        out = engine.affine_apply(half, log_s, t, reverse)
        ctx.reverse = reverse
        ctx.save_for_backward(out, log_s, t)
        return out, (-log_s if reverse else log_s)

    @staticmethod
    def backward(ctx, dout, dret):
        out, log_s, t = ctx.saved_tensors
        _, g_ls, g_t, din = engine.affine_backward(out, log_s, t, dout.float(), None if dret is None else dret.float(), ctx.reverse)
        return din, g_ls, g_t, None


class AffineCouplingBlock(Reversible):
    """Affine coupling z_a = x_a, z_b = x_b * exp(log_s) + t with (log_s, t) = F(x_a, y)  (efficient_modules.py:57-96).
    With `transform_type` = this package's WN the whole block is the fused HIP path (wg_coupling_*); any other transform module runs as
    it is and the block's own arithmetic goes through wg_affine_apply / wg_affine_backward (_GenericCoupling)."""

    def __init__(self, transform_type, memory_efficient=True, reverse_mode=False, **kwargs):
        super().__init__(reverse_mode)
        self.F = transform_type(**kwargs)
        self._memory_efficient = bool(memory_efficient)
        self._fn = _Coupling if hasattr(self.F, "hip_dims") else _GenericCoupling
        self._engine = None
        if self._fn is _Coupling:
            self._engine = engine.CouplingEngine(WgWnDims(*self.F.hip_dims(), _default_precision(), int(getattr(self.F, "has_bias", False))))

    def _run(self, x: Tensor, y: Tensor, reverse: bool) -> Tuple[Tensor, Tensor]:
        if self._fn is _GenericCoupling and not self._memory_efficient:
            # the graph is kept, as upstream :77-82 / :91-96: the transform under plain autograd, the map as its own node, x left alone
            ic = x.size(1) // 2
            xa, xb = x[:, :ic], x[:, ic:]
            log_s, t = self.F(xa, y)
            zb, ret = _AffineMap.apply(xb.contiguous(), log_s, t, reverse)
            return torch.cat((xa.to(zb.dtype), zb), 1), ret
        z, log_s = self._fn.apply(x, y, self, reverse, *self.F.parameters())
        if self._memory_efficient:
            _free(x)
        return z, log_s

    def forward_computation(self, x: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
        return self._run(x, y, False)

    def reverse_computation(self, z: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
        return self._run(z, y, True)
