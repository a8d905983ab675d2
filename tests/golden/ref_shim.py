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
    if "torchaudio" not in sys.modules:
        class MuLawEncoding(torch.nn.Module):
            def __init__(self, quantization_channels: int = 256) -> None:
                super().__init__()
                self.quantization_channels = quantization_channels

            def forward(self, x):
                mu = torch.tensor(self.quantization_channels - 1.0, dtype=x.dtype)
                x_mu = torch.sign(x) * torch.log1p(mu * torch.abs(x)) / torch.log1p(mu)
                return ((x_mu + 1) / 2 * mu + 0.5).to(torch.int64)

        ta = types.ModuleType("torchaudio")
        tr = types.ModuleType("torchaudio.transforms")
        tr.MuLawEncoding = MuLawEncoding
        ta.transforms = tr
        ta.__wg_standin__ = True
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.transforms"] = tr
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
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        tr = types.ModuleType("torchaudio.transforms")
        ta.transforms = tr
        ta.__wg_standin__ = True
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.transforms"] = tr
    tr = sys.modules["torchaudio.transforms"]
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
