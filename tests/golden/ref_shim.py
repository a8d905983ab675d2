"""Import shim for the upstream reference (ONLY usable in the build container).

`import model` of the reference fails here (pytorch_lightning / torchaudio / the
dataset submodule are absent), so an empty package `model` whose __path__ points at
/root/reference/model is registered and the hot-path submodules are imported one by
one.  Nothing from the reference is copied: this file only manipulates sys.modules.
Used by make_golden.py (fixture generation) and tests marked `reference`.
"""
import os
import sys
import types
import warnings

REF_ROOT = os.environ.get("WG_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


def load():
    """Returns a namespace with WaveGlow, WN, InvertibleConv1x1, AffineCouplingBlock, WaveGlowLoss."""
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    warnings.filterwarnings("ignore")
    if "model" not in sys.modules or getattr(sys.modules["model"], "__wg_shim__", False) is False:
        pkg = types.ModuleType("model")
        pkg.__path__ = [os.path.join(REF_ROOT, "model")]
        pkg.__wg_shim__ = True
        sys.modules["model"] = pkg
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)  # for the reference's top-level `utils`
    import importlib
    wg = importlib.import_module("model.waveglow")
    em = importlib.import_module("model.efficient_modules")
    ls = importlib.import_module("model.loss")
    ns = types.SimpleNamespace(
        WaveGlow=wg.WaveGlow, WN=wg.WN,
        InvertibleConv1x1=em.InvertibleConv1x1, AffineCouplingBlock=em.AffineCouplingBlock,
        WaveGlowLoss=ls.WaveGlowLoss)
    return ns


def _standin_transforms():
    """the stand-in `torchaudio.transforms` module (created once; every loader adds the class it restates)"""
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        tr = types.ModuleType("torchaudio.transforms")
        ta.transforms = tr
        ta.__wg_standin__ = True
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.transforms"] = tr
    return sys.modules["torchaudio.transforms"]


def load_wsrglow():
    """Returns the reference's WSRGlow class (model/wsrglow.py).

    wsrglow.py:4 imports `MuLawEncoding` from torchaudio, which is not installed in this image and is not part of
    /root/reference.  A stand-in module `torchaudio.transforms` exposing ONLY that class is registered for the import; it
    restates torchaudio's published algorithm (torchaudio.functional.mu_law_encoding):
        mu = quantization_channels - 1 ; x_mu = sign(x) * log1p(mu |x|) / log1p(mu) ; ((x_mu + 1) / 2 * mu + 0.5).to(int64)
    Everything else on the WSRGlow path (embeddings, torch.stft, AngleEmbedding, the WaveGlow flow stack) is the reference's own
    code, so the fixtures pin the path up to that one quantiser, which is pinned to the published formula only.
    """
    ns = load()
    import importlib
    import torch
    tr = _standin_transforms()
    if not hasattr(tr, "MuLawEncoding"):
        class MuLawEncoding(torch.nn.Module):
            def __init__(self, quantization_channels: int = 256) -> None:
                super().__init__()
                self.quantization_channels = quantization_channels

            def forward(self, x):
                mu = torch.tensor(self.quantization_channels - 1.0, dtype=x.dtype)
                x_mu = torch.sign(x) * torch.log1p(mu * torch.abs(x)) / torch.log1p(mu)
                return ((x_mu + 1) / 2 * mu + 0.5).to(torch.int64)

        tr.MuLawEncoding = MuLawEncoding
    ws = importlib.import_module("model.wsrglow")
    ns.WSRGlow = ws.WSRGlow
    return ns


def load_conditioners():
    """Returns the reference's LowPass / STFTDecimate classes (model/condition.py:22-66).

    Two things stand between that file and this image:
      * condition.py:4 imports torchaudio's MelSpectrogram (absent here; only MelSpec uses it) -- a placeholder class is registered
        that raises if it is ever instantiated;
      * LowPass.forward (condition.py:45-55) calls `torch.stft(x, nfft, hop, window=w)` / `torch.istft(real_view, ...)` with the
        pre-1.8 API: a real [..., 2] view in and out.  Current torch requires `return_complex=` and a complex istft input.  While the
        reference's forward runs, torch.stft / torch.istft are wrapped to supply exactly that legacy convention
        (return_complex=False; view_as_complex on the way into istft).  The arithmetic is torch's own stft / istft.
    """
    load()
    import contextlib
    import importlib
    import torch
    tr = _standin_transforms()
    if not hasattr(tr, "MelSpectrogram"):
        class MelSpectrogram(torch.nn.Module):                      # placeholder: never used by the fixtures
            def __init__(self, *a, **k):
                raise RuntimeError("torchaudio is not installed: MelSpectrogram is a placeholder")
        tr.MelSpectrogram = MelSpectrogram
    cond = importlib.import_module("model.condition")

    @contextlib.contextmanager
    def legacy_stft_api():
        stft, istft = torch.stft, torch.istft

        def stft_legacy(x, n_fft, hop_length=None, **kw):
            kw.setdefault("return_complex", False)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return stft(x, n_fft, hop_length, **kw)

        def istft_legacy(x, n_fft, hop_length=None, **kw):
            if not torch.is_complex(x):
                x = torch.view_as_complex(x.contiguous())
            return istft(x, n_fft, hop_length, **kw)

        torch.stft, torch.istft = stft_legacy, istft_legacy
        try:
            yield
        finally:
            torch.stft, torch.istft = stft, istft

    return types.SimpleNamespace(LowPass=cond.LowPass, STFTDecimate=cond.STFTDecimate, legacy_stft_api=legacy_stft_api)


def load_melspec():
    """Returns the reference's MelSpec class (model/condition.py:7-19) with a stand-in for torchaudio's MelSpectrogram.

    MelSpec = nn.ReflectionPad1d -> torchaudio.transforms.MelSpectrogram(center=False, ...) -> add_(1e-7).log_().  torchaudio is
    neither in /root/reference nor in this image, so a stand-in `torchaudio.transforms.MelSpectrogram` is registered for the import.
    It restates torchaudio's published algorithm with torch's own operators:
        Spectrogram:   torch.stft(x, n_fft, hop, win_length, window_fn(win_length) [periodic Hann], center, pad_mode,
                                  normalized=False, onesided=True, return_complex=True).abs().pow(power)
        MelScale:      fb = melscale_fbanks(n_fft // 2 + 1, f_min, f_max or sr // 2, n_mels, sr, norm=None, mel_scale="htk");
                       mel = (spec^T fb)^T
        melscale_fbanks (htk):  all_freqs = linspace(0, sr // 2, n_freqs); m_pts = linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2),
                       hz2mel(f) = 2595 log10(1 + f / 700); f_pts = 700 (10^(m / 2595) - 1);
                       fb = max(0, min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:])), slopes = f_pts[None] - all_freqs[:, None]
    So the fixtures pin the reflection pad and the log to the reference's own code and the STFT power to torch.stft; the mel
    filterbank above is held to transformers.audio_utils.mel_filter_bank (an independent implementation of the same htk definition
    that ships in this image) by make_golden.melspec_fixture, which also stores that matrix for the oracle's test.  The stand-in keeps the last power spectrogram in `.last_power` for the fixture.
    """
    load()
    import importlib
    import math
    import torch
    tr = _standin_transforms()

    class MelSpectrogram(torch.nn.Module):
        __wg_standin__ = True

        def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None, pad=0, n_mels=128,
                     window_fn=torch.hann_window, power=2.0, normalized=False, wkwargs=None, center=True, pad_mode="reflect",
                     onesided=None, norm=None, mel_scale="htk"):
            super().__init__()
            assert pad == 0 and not normalized and norm is None and mel_scale == "htk"
            self.n_fft = n_fft
            self.win_length = win_length if win_length is not None else n_fft
            self.hop_length = hop_length if hop_length is not None else self.win_length // 2
            self.power, self.center, self.pad_mode = power, center, pad_mode
            self.register_buffer("window", window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs))
            f_max = float(sample_rate // 2) if f_max is None else float(f_max)
            n_freqs = n_fft // 2 + 1
            all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
            m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
            m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
            m_pts = torch.linspace(m_min, m_max, n_mels + 2)
            f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
            f_diff = f_pts[1:] - f_pts[:-1]
            slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
            down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
            up = slopes[:, 2:] / f_diff[1:]
            self.register_buffer("fb", torch.max(torch.zeros(1), torch.min(down, up)))
            self.last_power = None

        def forward(self, x):
            shape = x.shape
            spec = torch.stft(x.reshape(-1, shape[-1]), self.n_fft, self.hop_length, self.win_length, self.window, center=self.center,
                              pad_mode=self.pad_mode, normalized=False, onesided=True, return_complex=True)
            spec = spec.abs().pow(self.power)
            spec = spec.reshape(shape[:-1] + spec.shape[-2:])
            self.last_power = spec.detach().clone()
            return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)

    if not getattr(getattr(tr, "MelSpectrogram", None), "__wg_standin__", False):
        tr.MelSpectrogram = MelSpectrogram
    sys.modules.pop("model.condition", None)                  # re-import so that condition.py binds the stand-in above
    cond = importlib.import_module("model.condition")
    return types.SimpleNamespace(MelSpec=cond.MelSpec)
