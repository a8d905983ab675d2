"""Negative log-likelihood of a flow output, the reference's model/loss.py:4-15, computed by the HIP kernels
wg_nll_loss / wg_nll_loss_backward."""
import torch
from torch.autograd import Function

from . import engine


class _NLL(Function):
    @staticmethod
    def forward(ctx, z, logdet, sigma, mean):
        ctx.save_for_backward(z)
        ctx.sigma, ctx.mean = sigma, mean
        return engine.nll_loss(z, logdet, sigma, mean)

    @staticmethod
    def backward(ctx, dloss):
        (z,) = ctx.saved_tensors
        dz, dlogdet = engine.nll_loss_backward(z, ctx.sigma, ctx.mean, dloss)
        return dz, dlogdet, None, None


class WaveGlowLoss(torch.nn.Module):
    """loss = mean_b(0.5 * sum_n z^2 / sigma^2 - logdet_b), divided by N when elementwise_mean."""

    def __init__(self, sigma=1., elementwise_mean=True):
        super().__init__()
        self.sigma2 = sigma ** 2
        self.mean = elementwise_mean

    def forward(self, z, logdet):
        if logdet.dim() == 0:                      # block-level callers pass the scalar T*logdet(W)
            logdet = logdet.expand(z.size(0))
        return _NLL.apply(z, logdet, self.sigma2 ** 0.5, self.mean)
