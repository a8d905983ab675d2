"""Conditioners of the reference's model/condition.py that sit on the training step.  MelSpec (condition.py:7-19) is the one every
WaveGlow / WaveFlow config uses; it runs as one HIP kernel (`wg_melspec`), which removes the torchaudio dependency of
`LightModel.training_step` (model/lightning.py:54).  torchaudio's MelSpectrogram defaults are restated (periodic Hann window, power
2, HTK mel scale, no filterbank norm); any other keyword raises WgError instead of being silently ignored."""
from torch import Tensor, nn

from . import engine
from ._lib import WgError

_DEFAULTS = dict(win_length=None, pad=0, power=2.0, normalized=False, center=False, onesided=None, norm=None, mel_scale="htk",
                 window_fn=None, wkwargs=None, pad_mode="reflect")


class MelSpec(nn.Module):
    def __init__(self, sr, n_fft, hop_length, **kwargs) -> None:
        super().__init__()
        self.sr, self.n_fft, self.hop_length = sr, n_fft, hop_length
        self.f_min = float(kwargs.pop("f_min", 0.0))
        self.f_max = kwargs.pop("f_max", None)
        self.n_mels = int(kwargs.pop("n_mels", 128))
        for k, v in kwargs.items():
            if k not in _DEFAULTS or (v != _DEFAULTS[k] and not (k == "win_length" and v == n_fft)):
                raise WgError("MelSpec(%s=%r) is not built into the HIP kernel (only torchaudio's defaults are)" % (k, v))

    def forward(self, x: Tensor) -> Tensor:
        return engine.melspec(x, self.sr, self.n_fft, self.hop_length, self.f_min, self.f_max, self.n_mels)


class LowPass(nn.Module):
    """STFT brick-wall low-pass (condition.py:22-57): `forward(x, i)` keeps the lowest `int((nfft//2+1) * ratio[i])` bins."""

    def __init__(self, nfft=1024, hop=256, ratio=(1 / 6, 1 / 3, 1 / 2, 2 / 3, 3 / 4, 4 / 5, 5 / 6, 1 / 1)):
        super().__init__()
        self.nfft, self.hop = nfft, hop
        self.cuts = [int((nfft // 2 + 1) * r) for r in ratio]

    def _run(self, x: Tensor, i: int, step: int) -> Tensor:
        shape = x.shape
        y = engine.lowpass(x.reshape(-1, shape[-1]), self.nfft, self.hop, self.cuts[int(i)], step)
        return y.view(*shape[:-1], y.size(-1))

    def forward(self, x: Tensor, r) -> Tensor:
        return self._run(x, r, 1)


class STFTDecimate(LowPass):
    """Low-pass to 1/r of the band, then keep every r-th sample (condition.py:60-66): the conditioner of configs/wsrglow_vctk_*.json."""

    def __init__(self, r, *args, **kwargs):
        super().__init__(*args, ratio=[1 / r], **kwargs)
        self.r = r

    def forward(self, x: Tensor) -> Tensor:
        return self._run(x, 0, self.r)
