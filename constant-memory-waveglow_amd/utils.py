"""Helpers the reference keeps in its top-level utils.py (utils.py:5-21): config-driven construction and the
old-style weight-norm toggles the modules of this package are parameterised with."""
import os

from torch import nn


def get_instance(module, config, *args):
    """`config` = {"type": <attribute of module>, "args": {...}} -> module.<type>(*args, **config["args"])."""
    factory = getattr(module, config["type"])
    return factory(*args, **config["args"])


def add_weight_norms(m):
    """For nn.Module.apply: put nn.utils.weight_norm (w = g * v/||v||, dim 0) on every module owning a `weight`."""
    if hasattr(m, "weight"):
        nn.utils.weight_norm(m)


def remove_weight_norms(m):
    """For nn.Module.apply: fold g, v back into a plain `weight` wherever weight norm is installed."""
    if hasattr(m, "weight_g"):
        nn.utils.remove_weight_norm(m)


def ensure_dir(path):
    os.makedirs(path, exist_ok=True)


def conv_gv(m):
    """(g, v) parameter pair of a conv for the C-ABI parameter table: (weight_g, weight_v) under weight norm,
    (None, weight) for a plain conv."""
    if hasattr(m, "weight_g"):
        return m.weight_g, m.weight_v
    return None, m.weight
