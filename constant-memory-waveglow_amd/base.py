"""Base classes of the flow modules.

Contract (what the reference's model/base.py:7-55 gives its callers, re-stated here in this package's own terms):

* a block or model implements two maps, `forward_computation` (data -> latent) and `reverse_computation` (latent -> data), each
  returning `(tensor, logdet-like)`;
* the public `forward` / `reverse` call those two, and swap them when the object was constructed with `reverse_mode=True` -- the
  switch the reference uses to train a model in its sampling direction;
* `FlowBase.infer(h, sigma)` samples a latent of the length the conditioning implies (`frames * hop`), maps it to data with
  autograd disabled and returns it squeezed; a 2-D conditioning tensor is treated as a batch of one.
"""
from typing import Callable, Tuple

import torch
from torch import Tensor, nn

Pair = Tuple[Tensor, Tensor]


class Reversible(nn.Module):
    """Two-directional module; `reverse_mode` decides which computation `forward` means."""

    def __init__(self, reverse_mode, **kwargs) -> None:
        super().__init__(**kwargs)
        self._reverse_mode = bool(reverse_mode)

    # -- the two maps a subclass provides ---------------------------------------------------------------------------------
    def forward_computation(self, x: Tensor, *args, **kwargs) -> Pair:
        raise NotImplementedError("%s does not define the data -> latent map" % type(self).__name__)

    def reverse_computation(self, z: Tensor, *args, **kwargs) -> Pair:
        raise NotImplementedError("%s does not define the latent -> data map" % type(self).__name__)

    # -- direction dispatch ---------------------------------------------------------------------------------------------------
    def _towards(self, latent: bool) -> Callable[..., Pair]:
        """The bound computation that maps towards the latent (True) or towards the data (False) as seen by the CALLER, i.e. with
        the reverse_mode swap applied."""
        if latent != self._reverse_mode:
            return self.forward_computation
        return self.reverse_computation

    def forward(self, x: Tensor, *args, **kwargs) -> Pair:
        return self._towards(True)(x, *args, **kwargs)

    def reverse(self, z: Tensor, *args, **kwargs) -> Pair:
        return self._towards(False)(z, *args, **kwargs)


class FlowBase(Reversible):
    """A conditional flow over audio: knows how many samples one conditioning frame stands for."""

    def __init__(self, condition_hop_length: int, reverse_mode=False) -> None:
        super().__init__(reverse_mode=reverse_mode)
        self._hop_length = int(condition_hop_length)

    def forward_computation(self, x: Tensor, h: Tensor) -> Pair:
        raise NotImplementedError("%s does not define the data -> latent map" % type(self).__name__)

    def reverse_computation(self, z: Tensor, h: Tensor) -> Pair:
        raise NotImplementedError("%s does not define the latent -> data map" % type(self).__name__)

    def infer(self, h: Tensor, sigma: float = 1.) -> Tensor:
        """Synthesis: h [n_mels, frames] or [B, n_mels, frames] -> audio with `frames * hop` samples per item (squeezed)."""
        cond = h[None] if h.dim() == 2 else h
        n_items, n_samples = cond.size(0), cond.size(2) * self._hop_length
        with torch.no_grad():
            latent = cond.new_empty((n_items, n_samples)).normal_(std=sigma)
            audio, _ = self._towards(False)(latent, cond)
        return audio.squeeze()
