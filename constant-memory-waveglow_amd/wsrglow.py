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


# front-end constants of the reference model (wsrglow.py:23-35); they are compiled into wg_wsr_cond as well
MU_LEVELS, MU_DIM = 256, 400            # mu-law quantiser levels, embedding width per low-rate sample
STFT_SIZE, STFT_HOP = 16, 8             # frame = 8 low-rate samples = one time step of the flow
PHASE_LEVELS, PHASE_DIM = 120, 50
N_BINS = STFT_SIZE // 2 + 1
COND_CHANNELS = STFT_HOP * MU_DIM + N_BINS * (1 + PHASE_DIM)     # 3659
assert COND_CHANNELS == engine.WSR_COND_CHANNELS


class WSRGlow(WaveGlow):
    """WaveGlow(12 flows, group = hop = 8 * rate, 2 early channels every 4 flows) conditioned on `_get_cond(low_rate_audio)`."""

    def __init__(self, upsample_rate: int = 2, memory_efficient: bool = False, **kwargs) -> None:
        group = STFT_HOP * upsample_rate
        super().__init__(flows=12, n_group=group, n_early_every=4, n_early_size=2, hop_size=group, n_mels=COND_CHANNELS,
                         memory_efficient=memory_efficient, **kwargs)
        self.mu_enc = nn.Sequential(MuLawEncoding(MU_LEVELS), nn.Embedding(MU_LEVELS, MU_DIM))
        self.angle_embed = AngleEmbedding(embed_num=PHASE_LEVELS, hidden_dim=PHASE_DIM)
        self.n_fft, self.hop_length = STFT_SIZE, STFT_HOP
        self.register_buffer("window", torch.hann_window(STFT_SIZE))       # kept for state-dict parity; the kernel builds its own

    def _get_cond(self, c):
        """low-rate audio [B, L] -> conditioning [B, 3659, L / 8]; clips `c` in place like the reference (wsrglow.py:38)."""
        engine.require_device(c)
        c.clamp_(-1.0, 1.0)
        return _WsrCondFn.apply(c, self.mu_enc[1].weight, self.angle_embed.embed.weight)

    def forward_computation(self, x, h):
        cond = self._get_cond(h)
        return WaveGlow.forward_computation(self, x, cond)

    def reverse_computation(self, z, h):
        cond = self._get_cond(h)
        return WaveGlow.reverse_computation(self, z, cond)
