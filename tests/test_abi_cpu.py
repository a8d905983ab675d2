"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/wgflow.h declares,
size queries behave, the torch modules mirror the reference's constructor / state-dict contract, and the product
path refuses to run without a HIP device (no CPU fallback).  No kernels are launched here."""
import ctypes as C
import os
import re

import pytest
import torch

import constant_memory_waveglow_amd as cm
from constant_memory_waveglow_amd import _lib, engine
import fill

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "wgflow.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wg_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _header_symbols()
    assert len(declared) >= 25
    for s in declared:
        assert hasattr(L, s), "libwgflow.so lacks %s" % s
    assert sorted(_lib.ABI_SYMBOLS) == declared
    assert L.wg_abi_version() == _lib.ABI_VERSION == 9
    header = open(os.path.join(ROOT, "include", "wgflow.h")).read()
    assert "#define WG_ABI_VERSION %d" % _lib.ABI_VERSION in header
    # the ctypes mirror of wg_config has exactly the fields the header declares, in order
    for cname, mirror in (("wg_config", _lib.WgConfig), ("wg_wn_dims", _lib.WgWnDims), ("wg_wf_config", _lib.WgWfConfig)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = [n.strip() for decl in re.findall(r"int32_t([^;]*);", body) for n in decl.split(",")]
        assert names == [n for n, _ in mirror._fields_], cname
    assert L.wg_strerror(0) == b"ok"


def test_size_queries_and_config_validation():
    L = _lib.lib()
    cfg = engine.make_config(12, 8, 4, 2, 256, 80, 256, 256, 256, 8, 3)
    assert L.wg_param_count(C.byref(cfg)) == 459                       # 459 state-dict entries
    assert (cfg.up_stride, cfg.up_kernel, cfg.up_pad) == (32, 65, 16)   # waveglow.py:125-129
    pk = L.wg_packed_bytes(C.byref(cfg))
    assert pk > 53_658_304 * 4                                          # every weight in >= 1 layout
    w0 = L.wg_workspace_bytes(C.byref(cfg), 24, 16000, 0)
    w1 = L.wg_workspace_bytes(C.byref(cfg), 24, 16000, 1)
    assert 0 < w0 < w1 < 8 << 30                                        # O(1) in flow depth: a few GB for ONE flow's activations
    cfg6 = engine.make_config(6, 8, 4, 2, 256, 80, 256, 256, 256, 8, 3)
    assert L.wg_workspace_bytes(C.byref(cfg6), 24, 16000, 1) <= w1     # does not grow with the number of flows
    assert L.wg_workspace_bytes(C.byref(cfg), 24, 16001, 1) == 0        # N % n_group != 0
    bad = engine.make_config(12, 8, 4, 2, 256, 80, 250, 256, 256, 8, 3)
    assert L.wg_packed_bytes(C.byref(bad)) == 0                         # channels must be multiples of 32


def test_state_dict_layout_matches_reference_names():
    cfg = fill.CONFIGS["c1"]
    m = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
    names = [n for n, _ in m.named_parameters()]
    specs = fill.model_param_specs(cfg)
    assert names == [n for n, _, _ in specs]
    sd = m.state_dict()
    for n, shape, _ in specs:
        assert tuple(sd[n].shape) == tuple(shape), n
    assert m.z_split_sizes == [2, 6]
    assert len(m.param_table()) == len(specs)
    # remove_weight_norm folds g,v into `weight` (inference.py:17 upstream)
    m.apply(cm.remove_weight_norms)
    assert "WNs.0.F.V.weight" in m.state_dict() and "WNs.0.F.V.weight_g" not in m.state_dict()
    tab = m.param_table()
    assert tab[1] is None and tab[2] is m.upsampler.weight


def test_bias_model_table_and_names():
    """WaveGlow(bias=True) (model/waveglow.py:58): the module tree carries the reference's bias parameters under the reference's names,
    and the C-ABI table appends each WN's biases behind its `end.weight` (wg_config.bias) in the order of fill.model_param_specs."""
    cfg = dict(fill.CONFIGS["micro_bias"])
    specs = fill.model_param_specs(cfg)
    cfg.pop("bias")
    m = cm.WaveGlow(memory_efficient=True, bias=True, **cfg)
    named = dict(m.named_parameters())
    assert sorted(named) == sorted(n for n, _, _ in specs)
    for n, shape, _ in specs:
        assert tuple(named[n].shape) == tuple(shape), n
    # the reference's own order interleaves the biases (nn.Conv1d registers weight before bias; weight norm re-registers g, v behind it)
    names = list(named)
    assert names.index("WNs.0.F.V.bias") < names.index("WNs.0.F.V.weight_g")
    tab = m.param_table()
    assert len(tab) == len(specs) == _lib.lib().wg_param_count(C.byref(m._engine.cfg))
    for t, (n, _, _) in zip(tab, specs):
        assert t is named[n], n
    assert float(m.WNs[0].F.end.bias.abs().sum()) == 0.0                 # zero_init covers the bias (waveglow.py:93-96)
    from constant_memory_waveglow_amd.parallel import waveglow_buckets
    ids = waveglow_buckets(cfg["flows"], cfg["depth"], bias=True)
    assert len(ids) == len(specs) and ids[-1] == cfg["flows"] - 1


def test_invconv_init_is_orthogonal_with_positive_det():
    blk = cm.InvertibleConv1x1(8)
    W = blk.weight.detach()[:, :, 0]
    assert torch.allclose(W @ W.t(), torch.eye(8), atol=1e-5)
    assert torch.det(W) > 0


def test_wn_zero_init_and_ctor_errors():
    wn = cm.WN(4, 80, 64, 64, 64, depth=3)
    assert float(wn.end.weight.abs().max()) == 0.0
    assert wn.r_field == 1 + 2 + 4 + 1
    wb = cm.WN(4, 80, 64, 64, 64, depth=3, bias=True)                  # built since round 3 (wg_config.bias)
    assert len(wb.param_table()) == 4 + 4 * 3 + 1 + 2 + 2 * 3 + 1 and wb.param_table()[-1] is wb.end.bias
    m = cm.WaveGlow(reverse_mode=True, memory_efficient=True, **fill.CONFIGS["micro"])
    assert m._engine.cfg.reverse_mode == 1 and m.z_split_sizes == [2, 6]


def test_no_cpu_fallback():
    m = cm.WaveGlow(memory_efficient=True, bias=False, **fill.CONFIGS["micro"])
    with pytest.raises(cm.WgError, match="no CPU fallback"):
        m(torch.rand(2, 512), torch.randn(2, 20, 8))
    with pytest.raises(cm.WgError):
        m.infer(torch.randn(20, 8))
    with pytest.raises(cm.WgError):
        cm.WaveGlowLoss(0.7)(torch.randn(2, 16), torch.zeros(2))
    with pytest.raises(cm.WgError):
        cm.InvertibleConv1x1(4)(torch.rand(1, 4, 16))


def test_shape_contract_assert():
    m = cm.WaveGlow(memory_efficient=True, bias=False, **fill.CONFIGS["micro"])
    with pytest.raises(AssertionError):                                  # mel shorter than the audio (waveglow.py:156)
        m(torch.rand(1, 512 * 4), torch.randn(1, 20, 8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "constant-memory-waveglow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("wg_oracle", "import oracle", "from oracle", "libwgoracle"):
                    assert needle not in text, (f, needle)


def test_hand_issued_loads_are_never_touched_before_their_wait():
    """The loader waves of convgemm16w_kernel issue global loads from inline asm and retire them with hand-counted waits.
    tools/check_asm_loads.py compiles the device code to ISA and walks every control-flow path of those kernels: no
    compiler-emitted instruction may read, copy or overwrite a staged register between its load and the wait that retires it."""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "asm loads" in l]
    # conv: 4 epilogues x 2 tile widths of convgemm16q + its three 256 x 128 (MG = 2) instantiations + 3 x convgemm16h; wgrad16s: 2 tile
    # heights + the paired launch; wgrad16t; the stage interpreter wf_rowsteps_kernel (three convgemm16h bodies inside); the one-launch
    # layers convlayer16h_kernel (two convgemm16h-shaped phases, the second with sc1 operand loads) and convlayer16q_kernel (the 256 x 128
    # form walking a list of gate and residual tiles); five 64-row (M64) and two column-group (CG2) instantiations of convgemm16q.  (The superseded
    # 32x32x16 kernels -- A/B builds only -- are checked with --defines WG_OPT_MFMA32.)
    # convgemm16g_kernel (LDS-DMA, no register-destination loads): the instruction counts behind its counted waits: four epilogues (gate, S-plane store, fp32 store, the K parts of a split product) and
    # the one-launch layer convlayer16g_kernel (two products one after the other)
    dma = [l for l in r.stdout.splitlines() if "LDS-DMA instructions" in l]
    assert len(dma) == 5 and any("convlayer16g_kernel" in l for l in dma) and all(l.rstrip().endswith(" 0 violations") for l in dma), r.stdout
    assert len(lines) == 39 and any("convlayer16h_kernel" in l for l in lines) and any("convlayer16q_kernel" in l for l in lines) and all(l.rstrip().endswith(" 0 violations") for l in lines), r.stdout


def test_wsrglow_state_dict_layout_matches_reference(golden_dir):
    """WSRGlow exposes the reference's names and shapes (model/wsrglow.py:21-35), recorded from the reference by
    make_golden.wsrglow_fixture: 173 parameters incl. mu_enc.1.weight / angle_embed.embed.weight, plus the `window` buffer."""
    import json
    import numpy as np
    gold = np.load(os.path.join(golden_dir, "model_wsr.npz"))
    want = [(k, tuple(shp)) for k, shp in json.loads(str(gold["state_dict_layout"]))]
    m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False, **fill.WSR_KW)
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == want
    assert m.n_group == 16 and m.n_mels == 3659 and m.z_split_sizes == [2, 2, 12]
    with pytest.raises(cm.WgError):
        m(torch.zeros(1, 1024), torch.zeros(1, 512))                   # CPU tensors: no fallback


def test_load_reference_checkpoint_strips_the_lightning_prefix():
    """A checkpoint written by the reference's LightModel keeps the flow under `model.` next to conditioner buffers
    (model/lightning.py:38-40); load_reference_checkpoint takes exactly that layout."""
    from constant_memory_waveglow_amd.parallel import load_reference_checkpoint
    cfg = fill.CONFIGS["micro"]
    src = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
    with torch.no_grad():
        for p in src.parameters():
            p.uniform_(-0.5, 0.5)
    ckpt = {"epoch": 3, "state_dict": {"model." + k: v.clone() for k, v in src.state_dict().items()}}
    ckpt["state_dict"]["conditioner.mel.1.spectrogram.window"] = torch.hann_window(1024)
    dst = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
    res = load_reference_checkpoint(dst, ckpt)
    assert not res.missing_keys and not res.unexpected_keys
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    load_reference_checkpoint(dst, src.state_dict())            # a bare state dict is accepted too


def test_param_table_follows_replaced_submodules():
    """The cached C-ABI parameter table (utils.SlotTable) must not keep handing out the parameters of a submodule that has been
    replaced, nor the original's parameters to a shallow copy of the model (an nn.DataParallel-style replica shares the cache object)."""
    import copy
    import torch
    import constant_memory_waveglow_amd as cm
    kw = dict(flows=2, n_group=8, n_early_every=4, n_early_size=2, hop_size=256, n_mels=20, memory_efficient=True, dilation_channels=32,
              residual_channels=32, skip_channels=32, depth=2, radix=3, bias=False)
    m = cm.WaveGlow(**kw)
    t0 = m.param_table()
    assert t0[0] is m.upsampler.bias and m.param_table()[0] is t0[0]
    # a parameter replaced in place (load_state_dict / .to()) stays behind the same slot
    m.upsampler.bias = torch.nn.Parameter(torch.ones_like(m.upsampler.bias))
    assert m.param_table()[0] is m.upsampler.bias
    # a replaced block: the table must follow
    other = cm.WaveGlow(**kw)
    old_end = m.WNs[1].F.end.weight
    m.WNs[1] = other.WNs[1]
    tab = m.param_table()
    assert tab[-1] is other.WNs[1].F.end.weight and tab[-1] is not old_end
    m.upsampler = other.upsampler
    assert m.param_table()[0] is other.upsampler.bias
    # weight norm removed: keys change, the table re-resolves (g slot -> None, v slot -> weight)
    m.apply(cm.remove_weight_norms)
    tab = m.param_table()
    assert tab[1] is None and tab[2] is m.upsampler.weight
    # a shallow copy resolves its own tree
    rep = copy.copy(m)
    rep._modules = dict(m._modules)
    rep._modules["upsampler"] = cm.WaveGlow(**kw).upsampler
    assert rep.param_table()[0] is rep._modules["upsampler"].bias
    assert m.param_table()[0] is m.upsampler.bias


def test_model_pickles_after_its_parameter_table_was_resolved():
    """torch.save(model), multiprocessing and ddp_spawn pickle whole modules; the cached table (a weak owner reference + the module
    tree's own dicts) must not ride along -- a restored or deep-copied model resolves its own."""
    import copy
    import io
    import pickle
    import torch
    import constant_memory_waveglow_amd as cm
    kw = dict(flows=2, n_group=8, n_early_every=4, n_early_size=2, hop_size=256, n_mels=20, memory_efficient=True, dilation_channels=32,
              residual_channels=32, skip_channels=32, depth=2, radix=3, bias=False)
    m = cm.WaveGlow(**kw)
    m.param_table()
    m2 = pickle.loads(pickle.dumps(m))
    t2 = m2.param_table()
    assert t2[0] is m2.upsampler.bias and t2[0] is not m.upsampler.bias and torch.equal(t2[0], m.upsampler.bias)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3.param_table()[-1] is m3.WNs[1].F.end.weight
    m.apply(cm.remove_weight_norms)       # (torch cannot deep-copy a module under the old-style weight norm: its `weight` is not a leaf)
    m.param_table()
    m4 = copy.deepcopy(m)
    assert m4.param_table()[0] is m4.upsampler.bias and m.param_table()[0] is m.upsampler.bias
    blk = cm.AffineCouplingBlock(cm.WN, True, in_channels=4, aux_channels=20, dilation_channels=32, residual_channels=32, skip_channels=32, depth=2)
    blk.F.param_table() if hasattr(blk.F, "param_table") else None
    pickle.loads(pickle.dumps(blk))


def test_packed_weights_key_is_committed_only_after_a_successful_pack():
    """engine.PackedWeights: a pack that fails half way must not leave a key behind under which a retry would skip the pack."""
    import torch
    from constant_memory_waveglow_amd import engine
    pw = engine.PackedWeights()
    params = [torch.zeros(3), None, torch.zeros(2)]
    pw.buf = torch.zeros(1)
    assert pw.stale(params) and pw.key is None            # nothing valid until commit()
    assert pw.stale(params)                               # the failed attempt left no key: still stale
    pw.commit()
    assert not pw.stale(params)
    params[0].add_(1.0)                                   # a parameter changed (version counter): stale again, and empty until re-packed
    assert pw.stale(params) and pw.key is None
