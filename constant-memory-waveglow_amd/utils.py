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


def conv_gv_slots(m):
    """conv_gv as (owner dict, key) pairs: where the two table entries of a conv live in the module tree right now.  A parameter
    replaced in place (`.to()`, `.half()`, load_state_dict, an optimizer step) stays behind the same slot; adding or removing weight
    norm changes the keys, which a stale slot shows as a KeyError (SlotTable below rebuilds then)."""
    if "weight_g" in m._parameters:
        return (m._parameters, "weight_g"), (m._parameters, "weight_v")
    return None, (m._parameters, "weight")


class SlotTable:
    """A module's C-ABI parameter table, resolved through cached (dict, key) slots instead of a walk over the module tree with
    nn.Module.__getattr__ on every call (0.5 ms for WaveGlow's 459 entries: a sixth of a single-utterance synthesis call).

    The cache is only as good as the tree it was resolved from, so every call re-checks it cheaply: the table belongs to ONE owner
    module (a shallow copy of the owner -- an nn.DataParallel-style replica -- shares this object through its __dict__ and gets its
    own resolution whenever it calls; the resolution is kept as one tuple, replaced as a unit), every parent -> child edge of the owner's tree must still hold the same child object (`model.WNs[k] = block`,
    `model.upsampler = ...` re-resolve; ~400 dict lookups, 30 us), and a KeyError from a slot (weight norm added or removed) re-resolves
    as before."""

    def __init__(self, method: str):
        # ONE attribute holds the whole resolution (weak owner reference, slots, edges): it is replaced as a unit, so a replica that
        # resolves on another thread can never leave this object with one owner's slots next to another owner's edges, and the table
        # does not keep its module alive (no module <-> table reference cycle).
        self._method, self._state = method, None

    # The cached resolution holds a weak reference and the module tree's own dicts: neither survives (nor belongs in) a pickle.  A
    # pickled / deep-copied table carries its method name only and resolves again on first use -- torch.save(model), multiprocessing
    # and ddp_spawn pickle whole modules (the reference's configs are trained through Lightning's ddp plugin, train.py:51-53).
    def __getstate__(self):
        return {"_method": self._method, "_state": None}

    def __setstate__(self, state):
        self._method, self._state = state["_method"], None

    def __deepcopy__(self, memo):
        return SlotTable(self._method)

    def _resolve(self, owner):
        import weakref
        slots = getattr(owner, self._method)()
        edges = [(m._modules, name, child) for m in owner.modules() for name, child in m._modules.items()]
        state = (weakref.ref(owner), slots, edges)
        self._state = state
        return state

    @staticmethod
    def _valid(state, owner):
        if state is None or state[0]() is not owner:
            return False
        for d, k, c in state[2]:
            if d.get(k) is not c:
                return False
        return True

    def __call__(self, owner):
        state = self._state                 # read once: everything below works on this snapshot
        if self._valid(state, owner):
            try:
                return [None if s is None else s[0][s[1]] for s in state[1]]
            except KeyError:                # weight norm was added or removed somewhere: resolve the tree again
                pass
        state = self._resolve(owner)
        return [None if s is None else s[0][s[1]] for s in state[1]]


def conv_gv(m):
    """(g, v) parameter pair of a conv for the C-ABI parameter table: (weight_g, weight_v) under weight norm,
    (None, weight) for a plain conv."""
    if hasattr(m, "weight_g"):
        return m.weight_g, m.weight_v
    return None, m.weight
