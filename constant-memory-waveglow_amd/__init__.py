"""constant-memory-waveglow_amd: the WaveGlow flow hot path of yoyololicon/constant-memory-waveglow on MI355X.

Exports mirror what the reference's `model` package exposes for this path (model/__init__.py:1-7):
WaveGlow, WSRGlow, FlowBase, Reversible, plus the block classes, WN and the NLL loss.  All computation happens in
csrc/libwgflow.so (hand-written HIP for gfx950) through the C ABI in include/wgflow.h.
"""
from .base import FlowBase, Reversible
from .efficient_modules import AffineCouplingBlock, InvertibleConv1x1
from .loss import WaveGlowLoss
from .utils import add_weight_norms, get_instance, remove_weight_norms
from .waveglow import WN, NonCausalLayer, WaveGlow, fused_gate
from .wsrglow import WSRGlow
from .waveflow import WaveFlow, WN2D
from .condition import LowPass, MelSpec, STFTDecimate
from ._lib import WgError

__all__ = ["WaveGlow", "WSRGlow", "WaveFlow", "WN2D", "MelSpec", "LowPass", "STFTDecimate", "WN", "NonCausalLayer", "fused_gate", "FlowBase", "Reversible", "InvertibleConv1x1",
           "AffineCouplingBlock", "WaveGlowLoss", "get_instance", "add_weight_norms", "remove_weight_norms", "WgError"]
