"""Direction-swapping bases, same contract as the reference's model/base.py:7-55:
`forward` runs the forward computation unless the module was built with reverse_mode=True (then the two
directions swap); FlowBase.infer draws z ~ N(0, sigma^2) and runs the inverse without autograd."""
from typing import Tuple

import torch
from torch import Tensor, nn


class Reversible(nn.Module):
    _reverse_mode: bool

    def __init__(self, reverse_mode, **kwargs) -> None:
        super().__init__(**kwargs)
        self._reverse_mode = reverse_mode

    def forward_computation(self, x: Tensor, *args, **kwargs) -> Tuple[Tensor, Tensor]:
        raise NotImplementedError

    def reverse_computation(self, z: Tensor, *args, **kwargs) -> Tuple[Tensor, Tensor]:
        raise NotImplementedError

    def forward(self, x: Tensor, *args, **kwargs) -> Tuple[Tensor, Tensor]:
        run = self.reverse_computation if self._reverse_mode else self.forward_computation
        return run(x, *args, **kwargs)

    def reverse(self, z: Tensor, *args, **kwargs) -> Tuple[Tensor, Tensor]:
        run = self.forward_computation if self._reverse_mode else self.reverse_computation
        return run(z, *args, **kwargs)


class FlowBase(Reversible):
    def __init__(self, condition_hop_length: int, reverse_mode=False) -> None:
        super().__init__(reverse_mode=reverse_mode)
        self._hop_length = condition_hop_length

    def forward_computation(self, x: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        raise NotImplementedError

    def reverse_computation(self, z: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        raise NotImplementedError

    @torch.no_grad()
    def infer(self, h: Tensor, sigma: float = 1.) -> Tensor:
        """h: [n_mels, frames] or [B, n_mels, frames] -> audio [B, frames*hop] (squeezed), base.py:42-55."""
        if h.dim() == 2:
            h = h.unsqueeze(0)
        batch, _, frames = h.shape
        z = h.new_empty((batch, frames * self._hop_length)).normal_(std=sigma)
        run = self.forward_computation if self._reverse_mode else self.reverse_computation
        x, _ = run(z, h)
        return x.squeeze()
