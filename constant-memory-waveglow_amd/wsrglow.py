"""WSRGlow (audio super-resolution) on the HIP flow engine: same constructor, parameter / buffer names and
forward/reverse contract as the reference's model/wsrglow.py.  The model is a WaveGlow with
n_group = hop = 8*upsample_rate and 3659 conditioning channels built from the low-rate signal by `_get_cond`:
mu-law embedding, 16-point STFT magnitude and phase embedding (wsrglow.py:37-50) -- one HIP kernel here
(`wg_wsr_cond`), with `wg_wsr_cond_backward` producing the two embedding-table gradients.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import engine
from .waveglow import WaveGlow


class MuLawEncoding(nn.Module):
    """Placeholder occupying slot 0 of `mu_enc` so that the embedding keeps its reference name `mu_enc.1.weight`
    (wsrglow.py:27-30; upstream this slot is torchaudio.transforms.MuLawEncoding, which has no parameters).
    The quantiser itself runs inside wg_wsr_cond."""

    def __init__(self, quantization_channels: int = 256):
        super().__init__()
        self.quantization_channels = quantization_channels

    def forward(self, x):
        raise engine.WgError("mu-law encoding is fused into the HIP conditioning kernel; call WSRGlow._get_cond")


class AngleEmbedding(nn.Module):
    """Parameter container of the phase embedding (wsrglow.py:8-18); evaluated inside wg_wsr_cond."""

    def __init__(self, embed_num, hidden_dim):
        super().__init__()
        self.embed_num = embed_num
        self.embed = nn.Embedding(num_embeddings=embed_num, embedding_dim=hidden_dim)

    def forward(self, index):
        raise engine.WgError("AngleEmbedding is fused into the HIP conditioning kernel; call WSRGlow._get_cond")


class _WsrCondFn(Function):
    @staticmethod
    def forward(ctx, c, mu_table, ang_table):
        ctx.save_for_backward(c)
        return engine.wsr_cond(c, mu_table.detach(), ang_table.detach())

    @staticmethod
    def backward(ctx, dcond):
        (c,) = ctx.saved_tensors
        dmu, dang = engine.wsr_cond_backward(c, dcond)
        return None, dmu, dang


class WSRGlow(WaveGlow):
    def __init__(self, upsample_rate: int = 2, memory_efficient: bool = False, **kwargs) -> None:
        super().__init__(12, 8 * upsample_rate, 4, 2, 8 * upsample_rate, 8 * 400 + 51 * 9,
                         memory_efficient=memory_efficient, **kwargs)
        self.mu_enc = nn.Sequential(MuLawEncoding(256), nn.Embedding(256, 400))
        self.angle_embed = AngleEmbedding(embed_num=120, hidden_dim=50)
        self.n_fft = 16
        self.hop_length = 8
        self.register_buffer('window', torch.hann_window(self.n_fft))

    def _get_cond(self, c):
        engine.require_device(c)
        c = c.clip_(-1, 1)                       # in place, as the reference (wsrglow.py:38)
        return _WsrCondFn.apply(c, self.mu_enc[1].weight, self.angle_embed.embed.weight)

    def forward_computation(self, x, h):
        return super().forward_computation(x, self._get_cond(h))

    def reverse_computation(self, z, h):
        return super().reverse_computation(z, self._get_cond(h))
