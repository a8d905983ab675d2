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
