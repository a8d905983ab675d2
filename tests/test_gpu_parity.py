"""Parity tests proper (-m gpu): the HIP engine, called through the C ABI by the torch modules, against the CPU
oracle on the same seeded inputs and against the golden vectors the upstream reference produced.

Tolerances (BASELINE.json north_star: 1e-4 fp32):  z / x abs 1e-4;  loss 1e-6;  logdet rtol 1e-4 (+1e-7 per sample:
the reference's own fp32 logdet sits ~1e-4 from fp64);  every parameter gradient within 1e-4 of
that tensor's max-abs.  Block-level tolerances follow the reference's tests/test_fwd_bwd.py."""
import os
import warnings

import numpy as np
import pytest
import torch

import fill
from make_golden import COUPLING_CASES
from oracle import wg_oracle as orc
import constant_memory_waveglow_amd as cm

pytestmark = pytest.mark.gpu

Z_ATOL, LOSS_ATOL, GRAD_RTOL = 1e-4, 1e-6, 1e-4


@pytest.fixture(params=["f32", "bf16x3", "bf16x3p"], autouse=True)
def precision(request, monkeypatch):
    """Every parity test runs in the three arithmetic modes of the contraction kernels (include/wgflow.h WG_PREC_*):
    exact fp32 MFMA, bf16x3 split on the fly, bf16x3 from pre-split S-planes (the default).  Same tolerances for all."""
    monkeypatch.setenv("WG_PRECISION", request.param)
    return request.param


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu suite needs the MI355X"
    from constant_memory_waveglow_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def npy(t):
    return t.detach().cpu().numpy()


def relmax(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-30))


def logdet_close(a, b, N):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-7 * N))


def _check_summary(gold, m, specs, z, logdet, loss, N):
    """A summary fixture of a TIMED workload (tests/golden/make_golden.py: _summary -- one step of the reference at a benchmarked size):
    both ends and the per-item norm of z, logdet, loss, norm and head of every gradient."""
    zz = npy(z)
    assert np.abs(zz[:, :256] - gold["z_head"]).max() < Z_ATOL and np.abs(zz[:, -256:] - gold["z_tail"]).max() < Z_ATOL
    zn = np.sqrt((zz.astype(np.float64) ** 2).sum(1))
    assert np.all(np.abs(zn - gold["z_item_norm"]) <= 1e-5 * gold["z_item_norm"])
    assert logdet_close(npy(logdet), gold["logdet"], N)
    assert abs(float(loss) - float(gold["loss"])) < LOSS_ATOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        if n.endswith("start.weight_v") and named[n].dim() == 4:
            continue                                             # WN2D's Conv2d(1, C, 1) under weight norm: exactly zero, rounding noise
        g = npy(named[n].grad)
        nh = min(g.size, gold["grad_head"].shape[1])
        assert np.abs(g.ravel()[:nh] - gold["grad_head"][i][:nh]).max() / max(float(gold["grad_max"][i]), 1e-30) < GRAD_RTOL, n
        gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        assert abs(gn - float(gold["grad_norm"][i])) <= 1e-4 * float(gold["grad_norm"][i]) + 1e-12, n


def build(name, dev, mem_eff=True, reverse_mode=False):
    cfg = fill.CONFIGS[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    kw = dict(cfg)
    m = cm.WaveGlow(memory_efficient=mem_eff, bias=kw.pop("bias", False), reverse_mode=reverse_mode, **kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    return m.to(dev), cfg, specs, P


@pytest.mark.parametrize("name", ["micro", "micro_bias", "micro_r5", "c1", "wsr_like"])
def test_model_step_vs_oracle_and_golden(dev, golden_dir, name):
    m, cfg, specs, P = build(name, dev)
    B, N, F = fill.SHAPES[name]
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True)
    gold = np.load(os.path.join(golden_dir, "model_%s.npz" % name))
    x, ht = T(audio, dev), T(h, dev).requires_grad_(True)
    z, logdet = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    for want in (ref, gold):
        assert np.abs(npy(z) - want["z"]).max() < Z_ATOL
        assert logdet_close(npy(logdet), want["logdet"], N)
        assert abs(float(loss) - float(want["loss"])) < LOSS_ATOL
        assert relmax(npy(ht.grad), want["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        g = npy(named[n].grad)
        assert relmax(g, ref["grads"][i]) < GRAD_RTOL, n
        nh = min(g.size, gold["grad_head"].shape[1])
        assert np.abs(g.ravel()[:nh] - gold["grad_head"][i][:nh]).max() / max(float(gold["grad_max"][i]), 1e-30) < GRAD_RTOL, n
        if "grad::" + n in gold:
            assert relmax(g, gold["grad::" + n]) < GRAD_RTOL, n
    assert torch.equal(x, T(audio, dev))                       # the caller's audio survives (waveglow.py:153 copies)


def test_wide_batch_step_vs_oracle(dev):
    """C1 at batch 9 (4 608 columns per launch): from 4 096 columns on the engine computes the skip sum and the conditioning gradient
    as ONE product per WN and carries the residual stream / its gradient as S-planes only (wgflow.hip: fused_skip, fused_dy,
    s_only_chain) -- code the small fixtures never reach.  Against the float64 oracle on the same inputs."""
    m, cfg, specs, P = build("c1", dev)
    B, (_, N, F) = 9, fill.SHAPES["c1"]
    audio, h = fill.inputs("c1x9", B, N, F, cfg["n_mels"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True, double=True)
    x, ht = T(audio, dev), T(h, dev).requires_grad_(True)
    z, logdet = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    assert abs(float(loss) - ref["loss"]) < LOSS_ATOL
    assert relmax(npy(ht.grad), ref["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n
    with torch.no_grad():
        xr, _ = m.reverse(z.detach(), ht.detach())
    assert np.abs(npy(xr) - audio).max() < Z_ATOL


def test_rank_form_of_the_skip_path_with_sixteen_rows_vs_oracle(dev, monkeypatch):
    """The skip path in its rank-2ic form (csrc/wgflow.hip lowrank_on: no skip sum, no dS; the end conv contracts the gate planes with
    W_end Wskip_l, the gate backward takes G as a K segment, W_o's skip rows and dW_end come from P_l = G gate_l^T) runs by default where
    2 ic <= 8; the instantiations for more rows -- `end_affine_kernel<32, false, 1>`, two row slices of `pgate_kernel` -- are forced here
    (WG_LOWRANK=2) on the scaled-down WSRGlow core (n_group 16: 2 ic = 16, 14) at 11 x 384 columns, against the float64 oracle."""
    monkeypatch.setenv("WG_LOWRANK", "2")
    cm._lib.lib().wg_reload_env()
    m, cfg, specs, P = build("wsr_like", dev)
    B, (_, N, F) = 11, fill.SHAPES["wsr_like"]
    audio, h = fill.inputs("wsr_like_x11", B, N, F, cfg["n_mels"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True, double=True)
    x, ht = T(audio, dev), T(h, dev).requires_grad_(True)
    z, logdet = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    assert abs(float(loss) - ref["loss"]) < LOSS_ATOL
    assert relmax(npy(ht.grad), ref["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n
    with torch.no_grad():
        xr, _ = m.reverse(z.detach(), ht.detach())
    assert np.abs(npy(xr) - audio).max() < Z_ATOL


@pytest.mark.parametrize("name", ["c1", "wsr_like"])
def test_start_conv_folded_into_the_first_layer_vs_conv_over_h0(dev, precision, monkeypatch, name):
    """WN.start has rank ic, so the first layer's dilated conv runs over xa through the composed weight W_0[kt] W_start (csrc/wgflow.hip
    start_fold_on, wg_small.h start_fold_kernel; model/waveglow.py:99, 41-43) -- by default, i.e. in every other parity test.  Here the
    same step with the fold switched off (WG_START_FOLD=0: the conv over h_0): both against the float64 oracle, the two against each other
    far inside the bars, and NOT bit for bit (the switch must reach the kernels)."""
    if precision != "bf16x3p":
        pytest.skip("the fold exists in the S-plane mode only")
    m, cfg, specs, P = build(name, dev)
    _, N, F = fill.SHAPES[name]
    B = 5
    audio, h = fill.inputs(name + "_fold_x5", B, N, F, cfg["n_mels"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True, double=True)
    got = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("WG_START_FOLD", sw)
        cm._lib.lib().wg_reload_env()
        m.zero_grad()
        x, ht = T(audio, dev), T(h, dev).requires_grad_(True)
        z, logdet = m(x, ht)
        loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
        loss.backward()
        assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL and logdet_close(npy(logdet), ref["logdet"], N)
        assert abs(float(loss) - ref["loss"]) < LOSS_ATOL and relmax(npy(ht.grad), ref["dh"]) < GRAD_RTOL
        named = dict(m.named_parameters())
        for i, (n, _, _) in enumerate(specs):
            assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n
        with torch.no_grad():
            xr, _ = m.reverse(z.detach(), ht.detach())
        assert np.abs(npy(xr) - audio).max() < Z_ATOL
        got[sw] = (z.detach().clone(), {n: q.grad.clone() for n, q in named.items()})
    z1, g1 = got["1"]
    z0, g0 = got["0"]
    assert float((z1 - z0).abs().max()) < 2e-5 and not torch.equal(z1, z0)
    for n in g1:
        assert relmax(npy(g1[n]), npy(g0[n])) < 2e-5, n


def test_bias_wide_batch_step_vs_oracle(dev):
    """WN(bias=True) (model/waveglow.py:58) at 4 608 columns per launch: the one-product skip sum (its bias rows: one per layer), the
    S-plane-only residual stream and the grouped weight-gradient launch with the ones segment.  Against the float64 oracle."""
    m, cfg, specs, P = build("micro_bias", dev)
    B, N, F = 9, 8 * 512, 64
    audio, h = fill.inputs("micro_bias_x9", B, N, F, cfg["n_mels"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True, double=True)
    x, ht = T(audio, dev), T(h, dev).requires_grad_(True)
    z, logdet = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    assert abs(float(loss) - ref["loss"]) < LOSS_ATOL
    assert relmax(npy(ht.grad), ref["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n
    with torch.no_grad():
        xr, _ = m.reverse(z.detach(), ht.detach())
    assert np.abs(npy(xr) - audio).max() < Z_ATOL


@pytest.mark.parametrize("name", ["micro", "micro_bias", "micro_r5", "c1"])
def test_model_inverse_and_infer(dev, golden_dir, name):
    m, cfg, specs, P = build(name, dev)
    B, N, F = fill.SHAPES[name]
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    gold = np.load(os.path.join(golden_dir, "model_%s.npz" % name))
    ht = T(h, dev)
    with torch.no_grad():
        x, ld = m.reverse(T(gold["z"], dev), ht)
        zlat = fill.normal(name + "/latent", (B, N), fill.SIGMA)
        xs, _ = m.reverse(T(zlat, dev), ht)
    assert np.abs(npy(x) - gold["x_inv"]).max() < Z_ATOL
    assert np.abs(npy(x) - audio).max() < Z_ATOL               # reverse(forward(x)) == x
    assert logdet_close(npy(ld), gold["logdet_inv"], N)
    assert np.abs(npy(xs) - gold["x_from_latent"]).max() < Z_ATOL * max(1.0, float(np.abs(gold["x_from_latent"]).max()))
    torch.manual_seed(0)
    y = m.infer(ht[0], sigma=0.6)                              # 2-D h is accepted (base.py:44-45)
    assert y.shape == (F * cfg["hop_size"],) and bool(torch.isfinite(y).all())


def test_reverse_mode_model(dev, golden_dir):
    """WaveGlow(reverse_mode=True): the architecture of SURVEY.md a14 (flows and early splits in the opposite order)."""
    m, cfg, specs, P = build("micro", dev, reverse_mode=True)
    B, N, F = fill.SHAPES["micro"]
    audio, h = fill.inputs("micro", B, N, F, cfg["n_mels"])
    gold = np.load(os.path.join(golden_dir, "model_micro_rm.npz"))
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True, reverse_mode=True)
    x, ht = T(audio, dev), T(h, dev).requires_grad_(True)
    z, logdet = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    for want in (ref, gold):
        assert np.abs(npy(z) - want["z"]).max() < Z_ATOL
        assert logdet_close(npy(logdet), want["logdet"], N)
        assert abs(float(loss) - float(want["loss"])) < LOSS_ATOL
        assert relmax(npy(ht.grad), want["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), gold["grad::" + n]) < GRAD_RTOL, n
    with torch.no_grad():
        xr, ldr = m.reverse(z.detach(), ht.detach())
    assert np.abs(npy(xr) - gold["x_inv"]).max() < Z_ATOL and np.abs(npy(xr) - audio).max() < Z_ATOL
    assert logdet_close(npy(ldr), gold["logdet_inv"], N)
    y = m.infer(ht.detach()[0], sigma=0.6)
    assert y.shape == (F * cfg["hop_size"],) and bool(torch.isfinite(y).all())


def test_c2_single_segment_vs_golden(dev, golden_dir):
    """BASELINE.json configs[1] network (256ch, 12 flows) on one 16000-sample segment against the reference's numbers."""
    m, cfg, specs, P = build("c2", dev)
    B, N, F = fill.SHAPES["c2"]
    audio, h = fill.inputs("c2", B, N, F, cfg["n_mels"])
    gold = np.load(os.path.join(golden_dir, "model_c2.npz"))
    x, ht = T(audio, dev), T(h, dev).requires_grad_(True)
    z, logdet = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - gold["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), gold["logdet"], N)
    assert abs(float(loss) - float(gold["loss"])) < LOSS_ATOL
    assert relmax(npy(ht.grad), gold["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        g = npy(named[n].grad).ravel()
        nh = min(g.size, gold["grad_head"].shape[1])
        scale = max(float(gold["grad_max"][i]), 1e-30)
        assert np.abs(g[:nh] - gold["grad_head"][i][:nh]).max() / scale < GRAD_RTOL, n
        nrm = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        assert abs(nrm - gold["grad_norm"][i]) <= 1e-4 * gold["grad_norm"][i] + 1e-12, n
    with torch.no_grad():
        xr, _ = m.reverse(z.detach(), ht.detach())
    assert np.abs(npy(xr) - gold["x_inv"]).max() < Z_ATOL


@pytest.mark.parametrize("name", ["micro", "c1"])
def test_upsampler_alone_vs_reference_golden(dev, golden_dir, name):
    """WaveGlow._upsample_h / wg_upsample on its own (waveglow.py:126-130,210-212) against the reference's output and the oracle."""
    m, cfg, specs, P = build(name, dev)
    B, N, F = fill.SHAPES[name]
    _, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    gold = np.load(os.path.join(golden_dir, "block_upsampler.npz"))[name]
    with torch.no_grad():
        y = npy(m._upsample_h(T(h, dev)))
    want = orc.upsample(orc.make_config(**cfg), P["upsampler.bias"], P["upsampler.weight_g"], P["upsampler.weight_v"], h, gold.shape[2])
    u = cfg["hop_size"] // cfg["n_group"]
    K = 2 * u + 1
    assert y.shape == gold.shape == (B, cfg["n_mels"], (F - 1) * u - 2 * (K // 2 - u // 2) + K)      # ConvTranspose1d output length
    assert np.abs(y - gold).max() < 2e-6 and np.abs(y - want).max() < 2e-6


def test_full_size_properties(dev):
    """BASELINE.json configs[1] at its full size (B=24): size-independent properties instead of an oracle run."""
    m, cfg, specs, P = build("c2", dev)
    B, N, F = 24, 16000, 63
    audio, h = fill.inputs("c2full", B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    crit = cm.WaveGlowLoss(fill.SIGMA)
    z, logdet = m(x, ht)
    loss = crit(z, logdet)
    loss.backward()
    g_full = {n: p.grad.clone() for n, p in m.named_parameters()}
    with torch.no_grad():
        xr, ldr = m.reverse(z.detach(), ht)
    assert float((xr - x).abs().max()) < Z_ATOL                                    # forward o reverse = id
    assert float((logdet.detach() + ldr).abs().max()) < 1e-4 * float(logdet.abs().max()) + 1e-2   # logdet_fwd = -logdet_rev
    # batch items are independent units: item 5 alone gives the same z / logdet
    with torch.no_grad():
        z1, ld1 = m(x[5:6].clone(), ht[5:6])
    assert float((z1 - z[5:6]).abs().max()) < 1e-5
    assert abs(float(ld1[0] - logdet[5])) < 1e-4 * abs(float(logdet[5])) + 1e-3
    # ... and inverts alone as well (a launch of fewer tiles than CUs takes the 64 x 64-tile conv kernel, wg_gemm16h.h)
    with torch.no_grad():
        xr1, _ = m.reverse(z[5:6].detach().clone(), ht[5:6])
    assert float((xr1 - x[5:6]).abs().max()) < Z_ATOL
    # linearity of the gradient in the batch: mean of two half-batch gradients == full-batch gradient (DP semantics)
    m.zero_grad()
    for sl in (slice(0, 12), slice(12, 24)):
        zz, ll = m(x[sl].clone(), ht[sl])
        (0.5 * crit(zz, ll)).backward()
    for n, p in m.named_parameters():
        a, b = p.grad, g_full[n]
        assert float((a - b).abs().max()) <= GRAD_RTOL * float(b.abs().max()) + 1e-12, n


@pytest.mark.parametrize("cname", ["d4", "last", "r5d3"])
def test_noncausal_layer_alone_vs_reference_golden(dev, golden_dir, precision, cname):
    """NonCausalLayer.forward on its own (model/waveglow.py:18-46, 41-46) through wg_layer_apply: against the reference's own output
    (block_layer.npz) and a plain torch fp32 evaluation of the same formulas; weight norm on and off, the last-layer form (no residual),
    a dilation that is not a power of two and radix 5."""
    if precision != "f32":
        pytest.skip("the stand-alone layer always runs the exact-fp32 kernels")
    import torch.nn.functional as Fn
    from make_golden import LAYER_CASES, layer_inputs
    C, Cd, Cs, radix, dil, last, wn, B, Tn = LAYER_CASES[cname]
    P, x, y = layer_inputs(cname)
    gold = np.load(os.path.join(golden_dir, "block_layer.npz"))
    m = cm.NonCausalLayer(dil, Cd, C, Cs, radix, False, last_layer=last)
    if wn:
        m.apply(cm.add_weight_norms)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    else:
        m.load_state_dict({"W.weight": torch.from_numpy(P["W.weight_v"]), "W_o.weight": torch.from_numpy(P["W_o.weight_v"])})
    m = m.to(dev)
    with torch.no_grad():
        res, skip = m(T(x, dev), T(y, dev))

    def eff(v, g):
        v = torch.from_numpy(v)
        return v if g is None else v * (torch.from_numpy(g) / v.flatten(1).norm(dim=1).view(-1, 1, 1))
    W, Wo = eff(P["W.weight_v"], P.get("W.weight_g")), eff(P["W_o.weight_v"], P.get("W_o.weight_g"))
    xt = torch.from_numpy(x)
    xy = Fn.conv1d(xt, W, padding=dil * (radix - 1) // 2, dilation=dil) + torch.from_numpy(y)
    o = Fn.conv1d(torch.tanh(xy[:, :Cd]) * torch.sigmoid(xy[:, Cd:]), Wo)
    want_skip = o if last else o[:, C:]
    assert np.abs(npy(skip) - gold[cname + "/skip"]).max() < 2e-5 and np.abs(npy(skip) - want_skip.numpy()).max() < 2e-5
    if last:
        assert res is None
    else:
        assert np.abs(npy(res) - gold[cname + "/res"]).max() < 2e-5 and np.abs(npy(res) - (o[:, :C] + xt).numpy()).max() < 2e-5
    with pytest.raises(cm.WgError):                              # shapes outside the kernels' set are refused, not approximated
        cm.NonCausalLayer(1, 24, 32, 32, 3, False).to(dev)(torch.zeros(1, 32, 64, device=dev), torch.zeros(1, 48, 64, device=dev))


@pytest.mark.parametrize("cname", ["hd2d4", "last"])
def test_noncausal_layer2d_alone_vs_reference_golden(dev, golden_dir, precision, cname):
    """NonCausalLayer2D.forward on its own (model/waveflow.py:14-51): the 3x3 conv dilated (h_dilation, dilation), causal along the height
    axis, as nine K segments with plane-row offsets; against the reference's own output (block_layer.npz, keys 2d_*)."""
    if precision != "f32":
        pytest.skip("the stand-alone layer always runs the exact-fp32 kernels")
    from make_golden import LAYER2D_CASES, layer2d_inputs
    C, Cd, Cs, hd, dil, last, wn, B, H, W = LAYER2D_CASES[cname]
    P, x, y = layer2d_inputs(cname)
    gold = np.load(os.path.join(golden_dir, "block_layer.npz"))
    m = cm.waveflow.NonCausalLayer2D(hd, dil, Cd, C, Cs, 3, False, last_layer=last)
    if wn:
        m.apply(cm.add_weight_norms)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    else:
        m.load_state_dict({"W.weight": torch.from_numpy(P["W.weight_v"]), "W_o.weight": torch.from_numpy(P["W_o.weight_v"])})
    m = m.to(dev)
    with torch.no_grad():
        res, skip = m(T(x, dev), T(y, dev))
    assert np.abs(npy(skip) - gold["2d_" + cname + "/skip"]).max() < 2e-5
    if last:
        assert res is None
    else:
        assert np.abs(npy(res) - gold["2d_" + cname + "/res"]).max() < 2e-5


def _layer_grads_vs_golden(m, wn, last, x, y, tag, key, gold, dev):
    from make_golden import layer_seeds
    xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                          # (no "runs without autograd" warning any more)
        res, skip = m(xt, yt)
    assert skip.requires_grad and (res is None) == bool(last)
    gr, gs = layer_seeds(tag, x.shape, tuple(skip.shape))
    sc = (skip * skip * T(gs, dev)).sum()
    if res is not None:
        sc = sc + (res * T(gr, dev)).sum()
    sc.backward()
    assert relmax(npy(xt.grad), gold[key + "/dx"]) < GRAD_RTOL and relmax(npy(yt.grad), gold[key + "/dy"]) < GRAD_RTOL
    for n, q in m.named_parameters():
        assert relmax(npy(q.grad), gold[key + "/grad::" + (n if wn else n + "_v")]) < GRAD_RTOL, n
    # an output nobody uses (d res absent), an input that needs no gradient
    m.zero_grad()
    res2, skip2 = m(T(x, dev), T(y, dev))
    skip2.sum().backward()
    assert all(q.grad is not None and bool(torch.isfinite(q.grad).all()) for q in m.parameters())


@pytest.mark.parametrize("cname", ["d4", "last", "r5d3"])
def test_noncausal_layer_alone_is_differentiable_vs_reference_golden(dev, golden_dir, precision, cname):
    """NonCausalLayer is an ordinary differentiable module upstream (model/waveglow.py:18-46): `res, skip = layer(x, y)` followed by any
    loss gives gradients for x, y, W and W_o (g and v under weight norm).  Here the call is an autograd node (engine.LayerFn) whose
    backward is wg_layer_backward; every gradient against what the reference's own autograd gave (block_layer.npz)."""
    if precision != "f32":
        pytest.skip("the stand-alone layer always runs the exact-fp32 kernels")
    from make_golden import LAYER_CASES, layer_inputs
    C, Cd, Cs, radix, dil, last, wn, B, Tn = LAYER_CASES[cname]
    P, x, y = layer_inputs(cname)
    gold = np.load(os.path.join(golden_dir, "block_layer.npz"))
    m = cm.NonCausalLayer(dil, Cd, C, Cs, radix, False, last_layer=last)
    if wn:
        m.apply(cm.add_weight_norms)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    else:
        m.load_state_dict({"W.weight": torch.from_numpy(P["W.weight_v"]), "W_o.weight": torch.from_numpy(P["W_o.weight_v"])})
    _layer_grads_vs_golden(m.to(dev), wn, last, x, y, "layer/" + cname, cname, gold, dev)


@pytest.mark.parametrize("cname", ["hd2d4", "last"])
def test_noncausal_layer2d_alone_is_differentiable_vs_reference_golden(dev, golden_dir, precision, cname):
    """The same for NonCausalLayer2D (model/waveflow.py:14-51): nine taps with plane-row offsets in the weight gradient and the data
    gradient, d y summed over the height axis the conditioning was broadcast over."""
    if precision != "f32":
        pytest.skip("the stand-alone layer always runs the exact-fp32 kernels")
    from make_golden import LAYER2D_CASES, layer2d_inputs
    C, Cd, Cs, hd, dil, last, wn, B, H, W = LAYER2D_CASES[cname]
    P, x, y = layer2d_inputs(cname)
    gold = np.load(os.path.join(golden_dir, "block_layer.npz"))
    m = cm.waveflow.NonCausalLayer2D(hd, dil, Cd, C, Cs, 3, False, last_layer=last)
    if wn:
        m.apply(cm.add_weight_norms)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    else:
        m.load_state_dict({"W.weight": torch.from_numpy(P["W.weight_v"]), "W_o.weight": torch.from_numpy(P["W_o.weight_v"])})
    _layer_grads_vs_golden(m.to(dev), wn, last, x, y, "layer2d/" + cname, "2d_" + cname, gold, dev)


@pytest.mark.parametrize("cname", ["d4", "last", "2d_hd2d4", "2d_last"])
def test_noncausal_layer_alone_with_biases_vs_torch(dev, precision, cname):
    """NonCausalLayer(bias=True) / NonCausalLayer2D(bias=True) called on their own (model/waveglow.py:18-46, model/waveflow.py:14-51 accept
    the flag): xy = W(x) + b_W + y and W_o(z) + b_o, so the biases are constants on y and on the outputs around wg_layer_apply, and their
    gradients are sums of d y / of the output gradients.  Outputs and every gradient (x, y, the weights, both biases) against a float64
    torch evaluation of the reference's formulas on the CPU, with and without autograd."""
    if precision != "f32":
        pytest.skip("the stand-alone layer always runs the exact-fp32 kernels")
    import torch.nn.functional as Fn
    from make_golden import LAYER2D_CASES, LAYER_CASES, layer2d_inputs, layer_inputs
    two_d = cname.startswith("2d_")
    if two_d:
        C, Cd, Cs, hd, dil, last, _, B, H, W = LAYER2D_CASES[cname[3:]]
        P, x, y = layer2d_inputs(cname[3:])
        m = cm.waveflow.NonCausalLayer2D(hd, dil, Cd, C, Cs, 3, True, last_layer=last)
    else:
        C, Cd, Cs, radix, dil, last, _, B, Tn = LAYER_CASES[cname]
        P, x, y = layer_inputs(cname)
        m = cm.NonCausalLayer(dil, Cd, C, Cs, radix, True, last_layer=last)
    R = Cs if last else C + Cs
    bW, bO = fill.normal("layerb/" + cname + "/bW", (2 * Cd,)) * 0.5, fill.normal("layerb/" + cname + "/bO", (R,)) * 0.5
    m.load_state_dict({"W.weight": torch.from_numpy(P["W.weight_v"]), "W_o.weight": torch.from_numpy(P["W_o.weight_v"]),
                       "W.bias": torch.from_numpy(bW), "W_o.bias": torch.from_numpy(bO)})
    m = m.to(dev)

    def ref(xr, yr, Wr, Or, bWr, bOr):
        if two_d:                                               # waveflow.py:41-51
            tmp = Fn.pad(xr, [dil, dil, 2 * hd, 0])
            xy = Fn.conv2d(tmp, Wr, bWr, dilation=(hd, dil)) + yr
            o = Fn.conv2d(torch.tanh(xy[:, :Cd]) * torch.sigmoid(xy[:, Cd:]), Or, bOr)
        else:                                                   # waveglow.py:41-46
            xy = Fn.conv1d(xr, Wr, bWr, padding=dil * (radix - 1) // 2, dilation=dil) + yr
            o = Fn.conv1d(torch.tanh(xy[:, :Cd]) * torch.sigmoid(xy[:, Cd:]), Or, bOr)
        return (None, o) if last else (o[:, :C] + xr, o[:, C:])
    leaves = [torch.from_numpy(a).double().requires_grad_(True) for a in (x, y, P["W.weight_v"], P["W_o.weight_v"], bW, bO)]
    want_res, want_skip = ref(*leaves)
    with torch.no_grad():
        res, skip = m(T(x, dev), T(y, dev))
    assert np.abs(npy(skip) - want_skip.detach().numpy()).max() < 2e-5
    assert (res is None) == bool(last) and (last or np.abs(npy(res) - want_res.detach().numpy()).max() < 2e-5)
    gs = fill.normal("layerb/" + cname + "/gs", tuple(skip.shape))
    gr = None if last else fill.normal("layerb/" + cname + "/gr", tuple(res.shape))
    sc = (want_skip * want_skip * torch.from_numpy(gs).double()).sum()
    if not last:
        sc = sc + (want_res * torch.from_numpy(gr).double()).sum()
    want = torch.autograd.grad(sc, leaves)
    xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
    res, skip = m(xt, yt)
    sc = (skip * skip * T(gs, dev)).sum()
    if not last:
        sc = sc + (res * T(gr, dev)).sum()
    sc.backward()
    got = [xt.grad, yt.grad, m.W.weight.grad, m.W_o.weight.grad, m.W.bias.grad, m.W_o.bias.grad]
    for nme, a, b in zip(("dx", "dy", "dW", "dW_o", "db_W", "db_o"), got, want):
        assert relmax(npy(a), b.numpy()) < GRAD_RTOL, nme


@pytest.mark.parametrize("name", ["micro", "c1", "c2"])
def test_one_launch_layer_vs_two_launches(dev, precision, monkeypatch, name):
    """convlayer16h_kernel (wg_layer16h.h): one launch per WN layer -- gate conv -> gate -> W_o -> residual / skip, model/waveglow.py:41-46 --
    wherever both products are small-grid launches (single-utterance synthesis).  The gate crosses workgroups inside the launch (sc1
    stores, an arrival counter per column tile, sc1 loads); the arithmetic is the two launches' own, so forward and inverse must agree with
    the two-launch path (WG_LAYER_FUSION=0) to rounding, repeat bit for bit, and the launch counter must show that the fused kernel ran."""
    if precision != "bf16x3p":
        pytest.skip("the one-launch layer exists in the S-plane mode only")
    from constant_memory_waveglow_amd import _lib
    m, cfg, specs, P = build(name, dev)
    _, N, F = fill.SHAPES[name]
    audio, h = fill.inputs(name, 1, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    depth = cfg.get("depth", 8)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("WG_LAYER_FUSION", fused)
        cm._lib.lib().wg_reload_env()
        before = _lib.lib().wg_stat_layer_launches()
        runs = []
        with torch.no_grad():
            for rep in range(4):                                  # repeated: the counters of a launch must be clean for the next one
                z, ld = m(x.clone(), ht)
                xr, ldr = m.reverse(z, ht)
                runs.append((z, ld, xr, ldr))
        torch.cuda.synchronize()
        n = _lib.lib().wg_stat_layer_launches() - before
        assert n == (4 * 2 * cfg["flows"] * depth if fused == "1" else 0), n
        for later in runs[1:]:                                    # a hand-off that read stale bytes would not repeat bit for bit
            for a, b in zip(runs[0], later):
                assert torch.equal(a, b)
        res[fused] = runs[0]
    # the two forms run the same MFMA sequences; the compiler contracts the gate's tanh * sigmoid arithmetic differently in the two
    # kernels, so they agree to rounding, not bit for bit
    z1, ld1, x1, lr1 = res["1"]
    z0, ld0, x0, lr0 = res["0"]
    assert float((z1 - z0).abs().max()) < 1e-5 * max(1.0, float(z0.abs().max())) and float((x1 - x0).abs().max()) < 1e-5
    assert float((ld1 - ld0).abs().max()) < 1e-6 * float(ld0.abs().max()) + 1e-5
    assert float((x1 - x).abs().max()) < Z_ATOL


@pytest.mark.parametrize("B", [16, 24])
def test_layer_launch_of_the_training_shapes_vs_two_launches(dev, precision, monkeypatch, B):
    """convlayer16q_kernel (wg_layer16q.h): a layer's gate conv and residual product as ONE persistent launch where the gate conv fills the
    chip in whole rounds of 256 x 128 tiles (batch 16: two rounds, batch 24: three -- the headline shape).  The gate crosses workgroups
    inside the launch (write-through stores, an arrival counter per column tile published half a tile late, sc1 loads).  A whole training
    step -- forward, recompute, backward -- must agree with the two-launch path (WG_LAYER_FUSION_BIG=0) to rounding, repeat bit for bit,
    and the launch counter must show the fused kernel ran in the forward and in the recompute pass of every flow."""
    if precision != "bf16x3p":
        pytest.skip("the one-launch layer exists in the S-plane mode only")
    from constant_memory_waveglow_amd import _lib
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    m, cfg, specs, P = build("c2", dev)
    N, F = 16000, 63
    audio, h = fill.inputs("c2fused%d" % B, B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    tr = FlowTrainer(m, fill.SIGMA)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("WG_LAYER_FUSION_BIG", fused)
        cm._lib.lib().wg_reload_env()
        before = _lib.lib().wg_stat_layer_launches()
        runs = []
        for rep in range(3):
            loss, z, logdet = tr.step(x, ht)
            runs.append((loss.clone(), z.clone(), logdet.clone(), tr.fg.flat.clone()))
        torch.cuda.synchronize()
        n = _lib.lib().wg_stat_layer_launches() - before
        # per step: 7 of 8 layers of every flow in the forward, and in the recompute of every flow but the one the forward kept
        assert n == (3 * (cfg["flows"] * 7 + (cfg["flows"] - 1) * 7) if fused == "1" else 0), n
        for later in runs[1:]:
            for a, b in zip(runs[0], later):
                assert torch.equal(a, b)
        res[fused] = runs[0]
    (l1, z1, d1, g1), (l0, z0, d0, g0) = res["1"], res["0"]
    assert abs(float(l1) - float(l0)) < 1e-7
    assert float((z1 - z0).abs().max()) < 1e-5 and float((d1 - d0).abs().max()) < 1e-5 * float(d0.abs().max()) + 1e-4
    for b in range(tr.n_buckets):                                 # every gradient bucket (one per flow, the upsampler's) to 1e-5 of its max
        s, e = tr.fg.bucket_ranges[b]
        assert float((g1[s:e] - g0[s:e]).abs().max()) <= 1e-5 * float(g0[s:e].abs().max()), b


def test_c2_full_batch_vs_oracle(dev, precision):
    """The HEADLINE shape itself -- BASELINE.json configs[1]: 256 channels, 12 flows, batch 24 x 16000 samples
    (configs/waveglow_LJ_speech.json:6-29), 48 000 columns per launch: wgrad16t's two-phase plan at K = 48 000, the 750-tile persistent
    walks, XCD rows, the one-product skip / conditioning gradient -- through FlowTrainer.step (wg_train_step, what bench.py times)
    against the float64 torch-CPU oracle (oracle/torch_cpu.py, pinned to the reference's goldens by tests/test_oracle_golden.py).
    Batch items are independent, so the oracle runs as one worker process per share of the batch (train_step_parallel).
    Bars: z 1e-4, logdet rtol 1e-4, loss 1e-6, every one of the 459 gradients and dh (from the same call: FlowTrainer.want_dh) within
    1e-4 of its tensor's max; then the same for memory_efficient=False, the other benchmarked form, against the same oracle run."""
    if precision != "bf16x3p":
        pytest.skip("the headline shape is checked in the default arithmetic (CPU oracle time)")
    from oracle import torch_cpu
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    m, cfg, specs, P = build("c2", dev)
    B, N, F = 24, 16000, 63
    audio, h = fill.inputs("c2full", B, N, F, cfg["n_mels"])
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count() or 8
    except Exception:                                            # noqa: BLE001
        cores = os.cpu_count() or 8
    workers = max(1, min(B, cores // 8))
    ref = torch_cpu.train_step_parallel(cfg, fill.table(specs, P), audio, h, fill.SIGMA, workers=workers, threads=max(1, min(8, cores // workers)),
                                        need_dh=True, double=True)
    x, ht = T(audio, dev), T(h, dev)
    # (model_c2_full.npz: one training step of the REFERENCE at this size on the CPU -- fp32, so looser than the float64 oracle, but the
    # reference itself: a summary of z, logdet, loss, every gradient's norm / head, dh; tests/golden/make_golden.py c2_full)
    gold_ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_c2_full.npz"))
    # both benchmarked forms of the step against the ONE oracle run: the constant-memory config (the headline) and the same network with
    # stored activations (memory_efficient=False, configs/waveglow_LJ_speech_fast.json: `other_models.waveglow_memory_efficient_false`)
    for mem_eff in (True, False):
        if not mem_eff:
            del tr
            torch.cuda.empty_cache()
            m, cfg, specs, P = build("c2", dev, mem_eff=False)
        tr = FlowTrainer(m, fill.SIGMA)
        tr.want_dh = True                                        # d loss / d h from the timed path itself (wg_train_step's dh output)
        parts0 = cm._lib.lib().wg_stat_gate_part_launches()
        loss, z, logdet = tr.step(x, ht)
        # the timed shape runs the rank-2ic form of the skip path with the gate convs writing their share of `out`: 8 per WN pass --
        # 12 forward + 11 recompute passes in the constant-memory form, 12 forward passes with stored activations
        assert cm._lib.lib().wg_stat_gate_part_launches() - parts0 == 8 * (23 if mem_eff else 12)
        assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
        assert logdet_close(npy(logdet), ref["logdet"], N)
        assert abs(float(loss) - ref["loss"]) < LOSS_ATOL
        named = dict(m.named_parameters())
        worst = 0.0
        for i, (n, _, _) in enumerate(specs):
            e = relmax(npy(named[n].grad).astype(np.float64), ref["grads"][i])
            worst = max(worst, e)
            assert e < GRAD_RTOL, (mem_eff, n)
        assert relmax(npy(tr.last_dh).astype(np.float64), ref["dh"]) < GRAD_RTOL, mem_eff
        _check_summary(gold_ref, m, specs, z, logdet, loss, N)                      # ... and against the reference's own step at this size
        dh = npy(tr.last_dh)
        assert np.abs(dh[:, :, :4] - gold_ref["dh_head"]).max() <= GRAD_RTOL * float(gold_ref["dh_max"])
        assert abs(float(np.sqrt((dh.astype(np.float64) ** 2).sum())) - float(gold_ref["dh_norm"])) <= 1e-4 * float(gold_ref["dh_norm"])
        print("headline shape vs float64 oracle (memory_efficient=%s): |dz| %.2e, worst gradient %.2e of its tensor's max (%d oracle workers)"
              % (mem_eff, float(np.abs(npy(z) - ref["z"]).max()), worst, workers))


@pytest.mark.parametrize("c", [2, 4, 8])
@pytest.mark.parametrize("batch", [1, 3])
@pytest.mark.parametrize("rev", [False, True])
def test_invconv_block(dev, golden_dir, c, batch, rev):
    """InvertibleConv1x1 as upstream's test_conv1x1_fwd_bwd: outputs, logdet sign, freed input, re-materialised input, grads."""
    gold = np.load(os.path.join(golden_dir, "block_invconv.npz"))
    Tn = 64 if batch == 1 else 200
    tag = "invconv/c%d_b%d_t%d" % (c, batch, Tn)
    k = tag + ("/rev" if rev else "/fwd")
    W = fill.orthogonal(tag + "/W", c)
    x = fill.uniform(tag + "/x", (batch, c, Tn))
    gz = fill.normal(tag + "/gz", (batch, c, Tn))
    outs = []
    for mem_eff in (False, True):
        blk = cm.InvertibleConv1x1(c, memory_efficient=mem_eff).to(dev)
        blk.weight.data.copy_(T(W, dev).unsqueeze(-1))
        xt = T(x, dev).requires_grad_(True)
        xin = xt.clone()
        y, ld = blk.reverse(xin) if rev else blk(xin)
        yrev = y.detach().clone()
        xinv, ld2 = (blk(yrev) if rev else blk.reverse(yrev))
        assert torch.equal(ld, -ld2)                                            # test_fwd_bwd.py:51
        if mem_eff:
            assert xin.untyped_storage().size() == 0 and yrev.untyped_storage().size() == 0   # :57-64
        ((y * T(gz, dev)).sum() + ld * 0.37).backward()
        assert np.abs(npy(xin) - x).max() < 1e-6                                # input re-materialised (:70)
        assert np.abs(npy(xinv) - x).max() < 2e-6                               # :72
        assert np.abs(npy(y) - gold[k + "/y"]).max() < 2e-6
        assert abs(float(ld) - float(gold[k + "/logdet"])) < 1e-4 * max(1.0, abs(float(gold[k + "/logdet"])))
        assert relmax(npy(xt.grad), gold[k + "/dx"]) < 1e-5
        assert relmax(npy(blk.weight.grad)[:, :, 0], gold[k + "/dW"]) < 1e-5
        outs.append((npy(y), npy(blk.weight.grad)))
    assert np.allclose(outs[0][0], outs[1][0]) and np.allclose(outs[0][1], outs[1][1], atol=5e-7, rtol=0)   # :78-79


def test_invconv_negative_det_is_nan(dev):
    W = fill.orthogonal("negdet", 4)
    W[:, 0] = -W[:, 0]
    blk = cm.InvertibleConv1x1(4).to(dev)
    blk.weight.data.copy_(T(W, dev).unsqueeze(-1))
    _, ld = blk(T(fill.uniform("negdet/x", (1, 4, 8)), dev))
    assert bool(torch.isnan(ld))                                                # efficient_modules.py:38


@pytest.mark.parametrize("cname", list(COUPLING_CASES))
@pytest.mark.parametrize("rev", [False, True])
def test_coupling_block(dev, golden_dir, cname, rev):
    """AffineCouplingBlock(WN) as upstream's test_affine_fwd_bwd, plus golden values from the reference."""
    gold = np.load(os.path.join(golden_dir, "block_coupling.npz"))
    cs = COUPLING_CASES[cname]
    tag = "coupling/" + cname
    k = tag + ("/rev" if rev else "/fwd")
    wn = dict(in_channels=cs["c"] // 2, aux_channels=cs["aux"], residual_channels=cs["wn"], dilation_channels=cs["wn"],
              skip_channels=cs["wn"], depth=cs["depth"], radix=3)
    specs = fill.wn_param_specs("F.", cs["c"] // 2, cs["aux"], cs["wn"], cs["wn"], cs["wn"], cs["depth"], 3)
    P = fill.fill_params(specs, tag + "/")
    x = fill.uniform(tag + "/x", (cs["B"], cs["c"], cs["T"]))
    y = fill.normal(tag + "/y", (cs["B"], cs["aux"], cs["T"]))
    gz = fill.normal(tag + "/gz", x.shape)
    gls = fill.normal(tag + "/gls", (cs["B"], cs["c"] // 2, cs["T"]))
    res = []
    for mem_eff in (False, True):
        blk = cm.AffineCouplingBlock(cm.WN, mem_eff, zero_init=False, **wn)
        blk.load_state_dict({n: torch.from_numpy(v) for n, v in P.items()})
        blk = blk.to(dev)
        xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
        xin = xt.clone()
        z, ls = blk.reverse(xin, yt) if rev else blk(xin, yt)
        zrev = z.detach().clone()
        xinv, ls2 = blk(zrev, yt.detach()) if rev else blk.reverse(zrev, yt.detach())
        assert torch.equal(ls, -ls2)                                            # test_fwd_bwd.py:131
        if mem_eff:
            assert xin.untyped_storage().size() == 0 and zrev.untyped_storage().size() == 0   # :137-144
            assert torch.equal(yt.detach(), T(y, dev))                          # h untouched (:145-146)
        ((z * T(gz, dev)).sum() + (ls * T(gls, dev)).sum()).backward()
        assert np.abs(npy(xin) - x).max() < 1e-5                                # re-materialised (:152)
        assert np.abs(npy(xinv) - x).max() < 1e-5                               # round trip (:154)
        assert np.abs(npy(z) - gold[k + "/z"]).max() < 1e-5
        assert np.abs(npy(ls) - gold[k + "/log_s"]).max() < 1e-5
        assert relmax(npy(xt.grad), gold[k + "/dx"]) < GRAD_RTOL
        assert relmax(npy(yt.grad), gold[k + "/dy"]) < GRAD_RTOL
        named = dict(blk.named_parameters())
        for i, (n, _, _) in enumerate(specs):
            g = npy(named[n].grad)
            nrm = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
            assert abs(nrm - gold[k + "/grad_norm"][i]) <= GRAD_RTOL * gold[k + "/grad_norm"][i] + 1e-12, n
            if k + "/grad::" + n in gold:
                assert relmax(g, gold[k + "/grad::" + n]) < GRAD_RTOL, n
        res.append([npy(named[n].grad) for n, _, _ in specs])
    for a, b in zip(*res):
        assert np.allclose(a, b)                                                # efficient == naive (:160)


@pytest.mark.parametrize("rev", [False, True])
def test_coupling_block_with_bias_vs_torch_cpu(dev, rev):
    """AffineCouplingBlock(WN, bias=True) (model/waveglow.py:58, efficient_modules.py:57-96) through the block-level entry points
    (wg_coupling_apply / wg_coupling_backward with wg_wn_dims.bias), both directions, against plain torch autograd on the CPU over
    oracle/torch_cpu.py's WN (itself pinned to the reference's bias golden by tests/test_oracle_golden.py)."""
    from oracle import torch_cpu
    ic, aux, C, depth, B, Tn = 3, 20, 32, 3, 2, 200
    wn = dict(in_channels=ic, aux_channels=aux, residual_channels=C, dilation_channels=C, skip_channels=C, depth=depth, radix=3)
    specs = fill.wn_param_specs("F.", ic, aux, C, C, C, depth, 3, bias=True)
    tag = "coupling/bias"
    P = fill.fill_params(specs, tag + "/")
    x = fill.uniform(tag + "/x", (B, 2 * ic, Tn))
    y = fill.normal(tag + "/y", (B, aux, Tn))
    gz = fill.normal(tag + "/gz", x.shape)
    gls = fill.normal(tag + "/gls", (B, ic, Tn))
    # reference: autograd over the restated WN
    pt = [torch.from_numpy(P[n]).requires_grad_(True) for n, _, _ in specs]
    xr, yr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(y).requires_grad_(True)
    log_s, t = torch_cpu._wn_forward(pt, xr[:, :ic], yr, depth, C, 3)
    if rev:
        zr, lsr = torch.cat((xr[:, :ic], (xr[:, ic:] - t) / torch.exp(log_s)), 1), -log_s      # efficient_modules.py:90-96
    else:
        zr, lsr = torch.cat((xr[:, :ic], xr[:, ic:] * torch.exp(log_s) + t), 1), log_s         # :77-88
    ((zr * torch.from_numpy(gz)).sum() + (lsr * torch.from_numpy(gls)).sum()).backward()
    blk = cm.AffineCouplingBlock(cm.WN, True, zero_init=False, bias=True, **wn)
    blk.load_state_dict({n: torch.from_numpy(v) for n, v in P.items()})
    blk = blk.to(dev)
    xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
    z, ls = blk.reverse(xt.clone(), yt) if rev else blk(xt.clone(), yt)
    ((z * T(gz, dev)).sum() + (ls * T(gls, dev)).sum()).backward()
    assert np.abs(npy(z) - zr.detach().numpy()).max() < 1e-5 and np.abs(npy(ls) - lsr.detach().numpy()).max() < 1e-5
    assert relmax(npy(xt.grad), xr.grad.numpy()) < GRAD_RTOL and relmax(npy(yt.grad), yr.grad.numpy()) < GRAD_RTOL
    named = dict(blk.named_parameters())
    for (n, _, _), ref in zip(specs, pt):
        assert relmax(npy(named[n].grad), ref.grad.numpy()) < GRAD_RTOL, n
    ls_raw, t_raw = blk.F(T(x[:, :ic].copy(), dev), T(y, dev))                                  # WN.forward on its own (wg_wn_apply)
    assert np.abs(npy(ls_raw) - log_s.detach().numpy()).max() < 1e-5 and np.abs(npy(t_raw) - t.detach().numpy()).max() < 1e-5


@pytest.mark.parametrize("bias", [False, True])
def test_wn_on_its_own_is_differentiable_vs_torch_cpu(dev, precision, bias):
    """WN is an ordinary differentiable module upstream (model/waveglow.py:49-105): `log_s, t = wn(x, y)` followed by any loss must give
    gradients for x, y and every parameter.  Here the call is an autograd node (waveglow._WNFn) whose backward is wg_coupling_backward
    seeded with (d log_s, d t); checked against plain torch autograd on the CPU over oracle/torch_cpu.py's WN."""
    from oracle import torch_cpu
    ic, aux, C, depth, B, Tn = 4, 20, 64, 4, 2, 333
    specs = fill.wn_param_specs("", ic, aux, C, C, C, depth, 3, bias=bias)
    tag = "wnalone/grad%d" % bias
    P = fill.fill_params(specs, tag + "/")
    x = fill.uniform(tag + "/x", (B, ic, Tn))
    y = fill.normal(tag + "/y", (B, aux, Tn))
    gls, gt = fill.normal(tag + "/gls", (B, ic, Tn)), fill.normal(tag + "/gt", (B, ic, Tn))
    pt = [torch.from_numpy(P[n]).requires_grad_(True) for n, _, _ in specs]
    xr, yr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(y).requires_grad_(True)
    ls_r, t_r = torch_cpu._wn_forward(pt, xr, yr, depth, C, 3)
    # a loss that is not linear in the outputs, so that the gradients handed to the node depend on them
    ((ls_r * torch.from_numpy(gls)).sum() + (t_r * t_r * torch.from_numpy(gt)).sum()).backward()
    wn = cm.WN(ic, aux, C, C, C, depth=depth, zero_init=False, bias=bias)
    wn.load_state_dict({n: torch.from_numpy(v) for n, v in P.items()})
    wn = wn.to(dev)
    xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                          # (the "runs without autograd" warning is gone)
        ls, t = wn(xt, yt)
    assert ls.requires_grad and t.requires_grad
    ((ls * T(gls, dev)).sum() + (t * t * T(gt, dev)).sum()).backward()
    assert np.abs(npy(ls) - ls_r.detach().numpy()).max() < 1e-5 and np.abs(npy(t) - t_r.detach().numpy()).max() < 1e-5
    assert relmax(npy(xt.grad), xr.grad.numpy()) < GRAD_RTOL and relmax(npy(yt.grad), yr.grad.numpy()) < GRAD_RTOL
    named = dict(wn.named_parameters())
    for (n, _, _), ref in zip(specs, pt):
        assert relmax(npy(named[n].grad), ref.grad.numpy()) < GRAD_RTOL, n
    # only t used, x not requiring a gradient: still a node (the parameters need theirs), and no gradient comes back for x
    wn.zero_grad()
    ls2, t2 = wn(T(x, dev), T(y, dev))
    t2.sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in wn.parameters())
    with torch.no_grad():
        ls3, t3 = wn(T(x, dev), T(y, dev))
    assert not ls3.requires_grad and torch.equal(ls3, ls2) and torch.equal(t3, t2)


class _TinyTransform(torch.nn.Module):
    """a transform that is not WN: (log_s, t) = chunk(tanh(conv3(x_a) + conv1(y)))"""

    def __init__(self, in_channels, aux_channels, hidden=24):
        super().__init__()
        self.a = torch.nn.Conv1d(in_channels, hidden, 3, padding=1)
        self.b = torch.nn.Conv1d(aux_channels, hidden, 1)
        self.o = torch.nn.Conv1d(hidden, 2 * in_channels, 1)

    def forward(self, xa, y):
        ls, t = self.o(torch.tanh(self.a(xa) + self.b(y))).chunk(2, 1)
        return 0.5 * torch.tanh(ls), t


@pytest.mark.parametrize("mem_eff", [True, False])
@pytest.mark.parametrize("rev", [False, True])
def test_coupling_block_with_any_transform_vs_plain_autograd(dev, rev, mem_eff):
    """AffineCouplingBlock takes any `transform_type` upstream (model/efficient_modules.py:58-62).  With a transform that is not this
    package's WN the module runs as it is and the block's own arithmetic goes through wg_affine_apply / wg_affine_backward
    (efficient_modules._GenericCoupling): outputs, the freed-and-rebuilt input and every gradient against the plain composition
    `zb = xb * exp(log_s) + t` (or its inverse) under torch autograd, both directions.  memory_efficient=False keeps the graph as
    upstream :77-82 / :91-96 does: the caller's x is neither freed nor rewritten (efficient_modules._AffineMap)."""
    ic, aux, B, Tn = 3, 10, 2, 157
    torch.manual_seed(3)
    blk = cm.AffineCouplingBlock(_TinyTransform, mem_eff, in_channels=ic, aux_channels=aux).to(dev)
    x = T(fill.uniform("anyF/x", (B, 2 * ic, Tn)), dev)
    y = T(fill.normal("anyF/y", (B, aux, Tn)), dev)
    gz, gls = T(fill.normal("anyF/gz", (B, 2 * ic, Tn)), dev), T(fill.normal("anyF/gls", (B, ic, Tn)), dev)
    # plain composition
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    ls, t = blk.F(xr[:, :ic], yr)
    if rev:
        zr, lsr = torch.cat((xr[:, :ic], (xr[:, ic:] - t) / torch.exp(ls)), 1), -ls
    else:
        zr, lsr = torch.cat((xr[:, :ic], xr[:, ic:] * torch.exp(ls) + t), 1), ls
    ((zr * gz).sum() + (lsr * gls).sum()).backward()
    want = {n: p.grad.clone() for n, p in blk.F.named_parameters()}
    blk.zero_grad()
    # the block
    xt, yt = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    xin = xt.clone()
    z, lo = blk.reverse(xin, yt) if rev else blk(xin, yt)
    if mem_eff:
        assert xin.untyped_storage().size() == 0                                # memory_efficient: the input was freed (:75 / :88)
    else:
        keep = xin.detach().clone()
    ((z * gz).sum() + (lo * gls).sum()).backward()
    if mem_eff:
        assert xin.untyped_storage().size() > 0 and float((xin - x).abs().max()) < 1e-5      # ... and rebuilt in place by the backward
    else:
        assert torch.equal(xin.detach(), keep)                                  # the caller's live input: never written
    assert float((z - zr).abs().max()) < 1e-5 and float((lo - lsr).abs().max()) < 1e-6
    assert relmax(npy(xt.grad), npy(xr.grad)) < 1e-5 and relmax(npy(yt.grad), npy(yr.grad)) < 1e-5
    for n, p in blk.F.named_parameters():
        assert relmax(npy(p.grad), npy(want[n])) < 1e-5, n


@pytest.mark.parametrize("mem_eff", [True, False])
def test_coupling_block_with_any_transform_under_autocast(dev, mem_eff):
    """A transform that returns half-precision (log_s, t) -- any module under torch.autocast -- must not make the block's own float32
    arithmetic refuse them (upstream wraps its Functions in custom_fwd / custom_bwd, efficient_modules.py:101,117): outputs are float32
    and close to the float32 run, and a backward goes through."""
    ic, aux, B, Tn = 3, 10, 2, 96
    torch.manual_seed(4)
    blk = cm.AffineCouplingBlock(_TinyTransform, mem_eff, in_channels=ic, aux_channels=aux).to(dev)
    x = T(fill.uniform("acF/x", (B, 2 * ic, Tn)), dev)
    y = T(fill.normal("acF/y", (B, aux, Tn)), dev)
    z32, ls32 = blk(x.clone(), y)
    xt = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        z, ls = blk(xt.clone(), y)
    assert z.dtype == torch.float32 and ls.dtype == torch.float32
    assert float((z - z32).abs().max()) < 2e-2 and float((ls - ls32).abs().max()) < 2e-2
    (z.sum() + ls.sum()).backward()
    assert xt.grad is not None and bool(torch.isfinite(xt.grad).all())
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in blk.F.parameters())


def test_coupling_block_on_shared_b_tiles_vs_oracle(dev, precision):
    """One coupling block at the shipped WN width (256 channels, so the gate conv has 512 rows) on 2 x 16 384 columns: 1 024 gate-conv
    tiles = 2 per CU, the shape at which the engine runs the 256 x 128 form of the conv kernel (two compute groups sharing one B image,
    convgemm16q_kernel<.., MG = 2>) -- checked against the oracle, forward and backward."""
    if precision != "bf16x3p":
        pytest.skip("the shared-B tile exists in the S-plane mode only")
    wn = dict(in_channels=4, aux_channels=80, residual_channels=256, dilation_channels=256, skip_channels=256, depth=2, radix=3)
    specs = fill.wn_param_specs("F.", 4, 80, 256, 256, 256, 2, 3)
    tag = "coupling/mg2"
    P = fill.fill_params(specs, tag + "/")
    B, Tn = 2, 16384
    x = fill.uniform(tag + "/x", (B, 8, Tn))
    y = fill.normal(tag + "/y", (B, 80, Tn))
    gz = fill.normal(tag + "/gz", x.shape)
    gls = fill.normal(tag + "/gls", (B, 4, Tn))
    z_ref, ls_ref = orc.coupling_apply(wn, fill.table(specs, P), x, y)
    ref = orc.coupling_backward(wn, fill.table(specs, P), z_ref, y, gz, gls)
    blk = cm.AffineCouplingBlock(cm.WN, True, zero_init=False, **wn)
    blk.load_state_dict({n: torch.from_numpy(v) for n, v in P.items()})
    blk = blk.to(dev)
    xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
    z, ls = blk(xt.clone(), yt)
    ((z * T(gz, dev)).sum() + (ls * T(gls, dev)).sum()).backward()
    assert np.abs(npy(z) - z_ref).max() < 1e-5 and np.abs(npy(ls) - ls_ref).max() < 1e-5
    got = dict(dx=npy(xt.grad), dy=npy(yt.grad))
    want = ref
    assert relmax(got["dx"], want["dx"]) < GRAD_RTOL and relmax(got["dy"], want["dy"]) < GRAD_RTOL
    named = dict(blk.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), want["grads"][i]) < GRAD_RTOL, n


@pytest.mark.gpu
@pytest.mark.parametrize("B,Tn,want_store", [(12, 2000, False), (1, 24400, False), (24, 2000, True)])
def test_coupling_block_on_flattened_column_tiles_vs_oracle(dev, precision, B, Tn, want_store):
    """convgemm16g_kernel (csrc/wg_gemm16g.h): 256 x 192 tiles over the FLATTENED (plane row, time) columns, eight multiplying waves fed by
    LDS-DMA.  One coupling block at the shipped WN (256 channels, depth 8: halo 128) against the oracle, forward and backward, at shapes
    that put the tile seams everywhere: 12 x 2048 padded columns = 128 column tiles that straddle plane rows (2048 = 10.67 tiles); one
    plane row of 24 400 steps whose last column tile is partial; 24 x 2000, the headline columns, where the 256-row data-gradient conv
    runs on the kernel as well (one tile per CU).  The library names the instantiation every timed launch ran (wg_timer_read_name)."""
    if precision != "bf16x3p":
        pytest.skip("the LDS-DMA kernel exists in the S-plane mode only")
    import ctypes as C
    from constant_memory_waveglow_amd import _lib
    wn = dict(in_channels=4, aux_channels=80, residual_channels=256, dilation_channels=256, skip_channels=256, depth=8, radix=3)
    specs = fill.wn_param_specs("F.", 4, 80, 256, 256, 256, 8, 3)
    tag = "coupling/g192/%d" % B
    P = fill.fill_params(specs, tag + "/")
    x = fill.uniform(tag + "/x", (B, 8, Tn))
    y = fill.normal(tag + "/y", (B, 80, Tn))
    gz = fill.normal(tag + "/gz", x.shape)
    gls = fill.normal(tag + "/gls", (B, 4, Tn))
    z_ref, ls_ref = orc.coupling_apply(wn, fill.table(specs, P), x, y)
    ref = orc.coupling_backward(wn, fill.table(specs, P), z_ref, y, gz, gls)
    blk = cm.AffineCouplingBlock(cm.WN, True, zero_init=False, **wn)
    blk.load_state_dict({n: torch.from_numpy(v) for n, v in P.items()})
    blk = blk.to(dev)
    L = _lib.lib()
    timer = L.wg_timer_create(-1, 4096)
    L.wg_timer_attach(timer)
    try:
        xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
        z, ls = blk(xt.clone(), yt)
        ((z * T(gz, dev)).sum() + (ls * T(gls, dev)).sum()).backward()
        torch.cuda.synchronize()
    finally:
        L.wg_timer_attach(None)
    n = L.wg_timer_count(timer)
    info = (C.c_longlong * (5 * n))()
    L.wg_timer_read_info(timer, info, n)
    nb = C.create_string_buffer(256)
    sites = {}
    for i in range(n):
        assert L.wg_timer_read_name(timer, i, nb, 256) >= 0
        sites.setdefault((int(info[5 * i]), nb.value.decode()), 0)
        sites[(int(info[5 * i]), nb.value.decode())] += 1
    L.wg_timer_destroy(timer)
    gate = {k[1]: v for k, v in sites.items() if k[0] == _lib.K_CONV_GATE}
    layer = sum(v for k, v in sites.items() if k[0] == _lib.K_LAYER and k[1] == "convlayer16g_kernel")
    # (the first layer's gate conv reads xa through the composed weight -- start_fold_on: 6 chunks of K -- and takes a 16x16x32 kernel)
    if want_store:          # 256 column tiles, one per CU: layers 1-6 of 8 run gate conv + residual product as ONE launch (convlayer16g_kernel)
        assert layer == 12 and gate.get("convgemm16g_kernel<EPI_GATE_SO>", 0) == 2 and sum(gate.values()) == 4, sites
    else:                   # layers 1-7 of 8, forward + the backward's recompute
        assert layer == 0 and gate.get("convgemm16g_kernel<EPI_GATE_SO>", 0) == 14 and sum(gate.values()) == 16, sites
    store = sum(v for k, v in sites.items() if k[0] == _lib.K_CONV_STORE and k[1] == "convgemm16g_kernel<EPI_STORE_SO>")
    assert (store >= 8) == want_store, sites                                     # the 8 data-gradient convs (K = 1536) of the backward
    assert np.abs(npy(z) - z_ref).max() < 1e-5 and np.abs(npy(ls) - ls_ref).max() < 1e-5
    got = dict(dx=npy(xt.grad), dy=npy(yt.grad))
    assert relmax(got["dx"], ref["dx"]) < GRAD_RTOL and relmax(got["dy"], ref["dy"]) < GRAD_RTOL
    named = dict(blk.named_parameters())
    for i, (n_, _, _) in enumerate(specs):
        assert relmax(npy(named[n_].grad), ref["grads"][i]) < GRAD_RTOL, n_


@pytest.mark.gpu
def test_layer_as_one_launch_on_flattened_tiles_vs_two_launches(dev, precision, monkeypatch):
    """convlayer16g_kernel (csrc/wg_gemm16g.h): a layer's gate conv and residual product in ONE launch -- a workgroup owns whole 192-column
    tiles, computes both 256-row gate tiles and then, from its own stores, the residual product.  A whole training step at the headline
    columns (24 x 2000) must agree with the two-launch path (WG_LAYER_G=0) to rounding and repeat bit for bit, and the counter must show
    the kernel ran: 6 of 8 layers of every flow in the forward and in the recompute of every flow but the one the forward kept."""
    if precision != "bf16x3p":
        pytest.skip("the LDS-DMA kernels exist in the S-plane mode only")
    from constant_memory_waveglow_amd import _lib
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    m, cfg, specs, P = build("c2", dev)
    B, N, F = 24, 16000, 63
    audio, h = fill.inputs("c2layerg", B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    tr = FlowTrainer(m, fill.SIGMA)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("WG_LAYER_G", fused)
        cm._lib.lib().wg_reload_env()
        before = _lib.lib().wg_stat_layerg_launches()
        runs = []
        for rep in range(2):
            loss, z, logdet = tr.step(x, ht)
            runs.append((loss.clone(), z.clone(), logdet.clone(), tr.fg.flat.clone()))
        torch.cuda.synchronize()
        n = _lib.lib().wg_stat_layerg_launches() - before
        # (6 of 8 layers: the last has no residual product, the first -- WN.start folded into its weight, start_fold_on -- is a gate product of
        # 6 chunks and takes the 16x16x32 kernel + its own residual launch)
        assert n == (2 * (cfg["flows"] * 6 + (cfg["flows"] - 1) * 6) if fused == "1" else 0), n
        for a, b in zip(runs[0], runs[1]):
            assert torch.equal(a, b)
        res[fused] = runs[0]
    (l1, z1, d1, g1), (l0, z0, d0, g0) = res["1"], res["0"]
    # the same arithmetic in both forms (the residual product's K walk and accumulate-into value are unchanged): equal to the last bit
    # of fp32 accumulation order, which IS the same; allow rounding in case a shape takes another tile order
    assert abs(float(l1) - float(l0)) < 1e-6 and float((z1 - z0).abs().max()) < 2e-5 and float((d1 - d0).abs().max()) < 1e-2
    assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max())


def test_inverse_seam_launch_vs_three_launches(dev, precision, monkeypatch):
    """Synthesis between two WNs: the inverse affine, the inverse 1x1 conv and the next WN's start conv as ONE launch
    (end_affine_kernel<8, true>, opt-in WG_INV_SEAM=1: measured no faster, csrc/wgflow.hip run_inv_seam) against the three launches: the
    same expressions in the same order, so x must be equal bit for bit (logdet: the per-block sums of log_s are added in another order)."""
    if precision != "bf16x3p":
        pytest.skip("the seam kernel writes S-planes: the default arithmetic only")
    m, cfg, specs, P = build("c2", dev)
    for B, F in ((1, 63), (3, 20)):
        h = T(fill.normal("seam/h%d" % B, (B, cfg["n_mels"], F)), dev)
        z = T(fill.normal("seam/z%d" % B, (B, F * 256), 0.6), dev)
        out = {}
        for sw in ("1", "0"):
            monkeypatch.setenv("WG_INV_SEAM", sw)
            cm._lib.lib().wg_reload_env()
            with torch.no_grad():
                out[sw] = m.reverse(z.clone(), h)
        assert torch.equal(out["1"][0], out["0"][0])
        assert float((out["1"][1] - out["0"][1]).abs().max()) <= 1e-6 * float(out["0"][1].abs().max())


def test_wn_forward_standalone(dev):
    wn = cm.WN(4, 80, 64, 64, 64, depth=4, zero_init=False).to(dev)
    specs = fill.wn_param_specs("", 4, 80, 64, 64, 64, 4, 3)
    P = fill.fill_params(specs, "wnalone/")
    wn.load_state_dict({n: torch.from_numpy(v) for n, v in P.items()})
    x = fill.uniform("wnalone/x", (2, 4, 333))
    y = fill.normal("wnalone/y", (2, 80, 333))
    with torch.no_grad():
        ls, t = wn(T(x, dev), T(y, dev))
    xx = np.concatenate([x, np.zeros_like(x)], 1)
    z, lso = orc.coupling_apply(dict(in_channels=4, aux_channels=80, residual_channels=64, dilation_channels=64,
                                     skip_channels=64, depth=4, radix=3), fill.table(specs, P), xx, y)
    assert np.abs(npy(ls) - lso).max() < 1e-5
    assert np.abs(npy(t) - z[:, 4:]).max() < 1e-5                               # xb = 0  =>  zb = t


def test_weight_norm_removed_model_matches(dev):
    """inference.py:17 upstream: model.apply(remove_weight_norms) must not change the function."""
    m, cfg, specs, P = build("micro", dev)
    B, N, F = fill.SHAPES["micro"]
    audio, h = fill.inputs("micro", B, N, F, cfg["n_mels"])
    with torch.no_grad():
        z0, l0 = m(T(audio, dev), T(h, dev))
        m.apply(cm.remove_weight_norms)
        z1, l1 = m(T(audio, dev), T(h, dev))
    assert float((z0 - z1).abs().max()) < 1e-5 and float((l0 - l1).abs().max()) < 1e-3


def test_inverse_graph_replay_matches_eager(dev, monkeypatch):
    """WG_GRAPHS=1: wg_inverse captured into a hipGraph and replayed gives the same audio as the eager launches."""
    m, cfg, specs, P = build("micro", dev)
    B, N, F = fill.SHAPES["micro"]
    _, h = fill.inputs("micro", B, N, F, cfg["n_mels"])
    z1, z2 = T(fill.normal("g/z1", (B, N)), dev), T(fill.normal("g/z2", (B, N)), dev)
    with torch.no_grad():
        want = [m.reverse(z, T(h, dev))[0].clone() for z in (z1, z2)]
        monkeypatch.setenv("WG_GRAPHS", "1")
        got = [m.reverse(z, T(h, dev))[0].clone() for z in (z1, z2, z1)]      # capture, replay, replay
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[0])
    # the default (auto): a small call is captured on its third occurrence and replayed from then on
    m2, _, _, _ = build("micro", dev)
    monkeypatch.delenv("WG_GRAPHS", raising=False)
    with torch.no_grad():
        auto = [m2.reverse(z, T(h, dev))[0].clone() for z in (z1, z2, z1, z2, z1)]
    assert len(m2._engine._graphs) == 1
    assert all(torch.equal(a, want[i % 2]) for i, a in enumerate(auto))
    monkeypatch.setenv("WG_GRAPHS", "0")
    m3, _, _, _ = build("micro", dev)
    with torch.no_grad():
        for z in (z1, z2, z1, z2):
            m3.reverse(z, T(h, dev))
    assert len(m3._engine._graphs) == 0


def test_loss_kernel(dev):
    z = fill.normal("loss/z", (3, 4000))
    ld = fill.normal("loss/ld", (3,))
    zt, lt = T(z, dev).requires_grad_(True), T(ld, dev).requires_grad_(True)
    for mean in (True, False):
        zt.grad = lt.grad = None
        loss = cm.WaveGlowLoss(0.7, elementwise_mean=mean)(zt, lt)
        loss.backward()
        want = float(np.mean(0.5 * (z.astype(np.float64) ** 2).sum(1) / 0.49 - ld)) / (4000 if mean else 1)
        assert abs(float(loss) - want) < 1e-5 * max(1.0, abs(want))
        sc = 1.0 / 3 / (4000 if mean else 1)
        assert np.allclose(npy(zt.grad), z / 0.49 * sc, rtol=1e-5, atol=1e-9)
        assert np.allclose(npy(lt.grad), -sc, rtol=1e-6)


def test_trainer_step_matches_autograd(dev):
    """FlowTrainer (the bench's step: engine calls + flat gradient buffer) == the autograd path."""
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    m, cfg, specs, P = build("c1", dev)
    B, N, F = fill.SHAPES["c1"]
    audio, h = fill.inputs("c1", B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    z, logdet = m(x, ht)
    cm.WaveGlowLoss(fill.SIGMA)(z, logdet).backward()
    ga = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    loss, z2, ld2 = FlowTrainer(m, fill.SIGMA).step(x, ht)
    assert torch.equal(z, z2) and torch.equal(logdet, ld2)
    for a, p in zip(ga, m.parameters()):
        assert torch.equal(a, p.grad)


# ---- memory_efficient=False: stored-activation mode (configs/waveglow_LJ_speech_fast.json) --------------------------------

@pytest.mark.parametrize("name,rev", [("micro", False), ("c1", False), ("wsr_like", False), ("c1", True)])
def test_stored_activation_mode_vs_oracle(dev, name, rev):
    """WaveGlow(memory_efficient=False): the forward leaves every flow's WN layers in the workspace and the backward reads
    them (wg_config.keep_activations).  Same oracle, same tolerances as the constant-memory path; z / logdet are the same
    kernels on the same inputs, hence bit-identical to it."""
    m, cfg, specs, P = build(name, dev, mem_eff=False, reverse_mode=rev)
    m_ce = build(name, dev, mem_eff=True, reverse_mode=rev)[0]
    B, N, F = fill.SHAPES[name]
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True, reverse_mode=rev)
    x, ht = T(audio, dev).requires_grad_(True), T(h, dev).requires_grad_(True)
    z, logdet = m(x, ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    gen = m._engine._kept_gen
    loss.backward()
    assert m._engine._kept_gen == gen + 1                      # the stored activations were the ones used (and are spent now)
    z_ce, ld_ce = m_ce(T(audio, dev), T(h, dev))
    assert torch.equal(z, z_ce) and torch.equal(logdet, ld_ce)
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    assert abs(float(loss) - float(ref["loss"])) < LOSS_ATOL
    assert relmax(npy(ht.grad), ref["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n
    # dx: compare with the constant-memory path (the oracle's train_step does not return it)
    x2, h2 = T(audio, dev).requires_grad_(True), T(h, dev)
    z2, ld2 = m_ce(x2, h2)
    cm.WaveGlowLoss(fill.SIGMA)(z2, ld2).backward()
    assert relmax(npy(x.grad), npy(x2.grad)) < GRAD_RTOL


def test_stored_activations_overwritten_fall_back_to_recompute(dev):
    """Two forwards, then backward through the FIRST: its stored activations are gone, so the backward must notice and
    recompute (never read the second call's activations)."""
    m, cfg, specs, P = build("c1", dev, mem_eff=False)
    B, N, F = fill.SHAPES["c1"]
    audio, h = fill.inputs("c1", B, N, F, cfg["n_mels"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA)
    x, ht = T(audio, dev), T(h, dev)
    z, logdet = m(x, ht)
    m(T(audio[::-1] * 0.5, dev), T(h[::-1] + 0.25, dev))      # overwrites the workspace
    gen = m._engine._kept_gen
    cm.WaveGlowLoss(fill.SIGMA)(z, logdet).backward()
    assert m._engine._kept_gen == gen                          # recompute path taken
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n


def test_trainer_step_stored_activation_mode(dev):
    """FlowTrainer on a memory_efficient=False model (wg_train_step with keep_activations) == that model's autograd path,
    and within tolerance of the constant-memory trainer."""
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    m, cfg, specs, P = build("c1", dev, mem_eff=False)
    m_ce = build("c1", dev, mem_eff=True)[0]
    B, N, F = fill.SHAPES["c1"]
    audio, h = fill.inputs("c1", B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    z, logdet = m(x, ht)
    cm.WaveGlowLoss(fill.SIGMA)(z, logdet).backward()
    ga = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    loss, z2, ld2 = FlowTrainer(m, fill.SIGMA).step(x, ht)
    assert torch.equal(z, z2) and torch.equal(logdet, ld2)
    for a, p in zip(ga, m.parameters()):
        assert torch.equal(a, p.grad)
    loss_ce, _, _ = FlowTrainer(m_ce, fill.SIGMA).step(x, ht)
    assert abs(float(loss) - float(loss_ce)) < LOSS_ATOL
    for (n, p), q in zip(m.named_parameters(), m_ce.parameters()):
        if n.endswith("start.weight_v") and p.shape[1] == 1:
            continue                                           # fan-in 1: the true gradient is zero, what is left is rounding noise
        assert relmax(npy(p.grad), npy(q.grad)) < GRAD_RTOL, n


# ---- WSRGlow: conditioning front-end kernels and the model ----------------------------------------

def _wsr_tables():
    t = fill.wsr_tables("wsr/")
    return t["mu_enc.1.weight"], t["angle_embed.embed.weight"]


@pytest.mark.parametrize("B,L", [(1, 8), (2, 512), (3, 1208)])
def test_wsr_cond_kernel_vs_oracle(dev, B, L):
    from constant_memory_waveglow_amd import engine
    mu_w, ang_w = _wsr_tables()
    c = fill.uniform("wsrk/c%d_%d" % (B, L), (B, L), -1.3, 1.3)
    c[0, : min(L, 24)] = 0.0                                   # silence: atan2(0, 0), mu-law level 128
    if L >= 64:
        c[-1, 40:48] = 1.0
        c[-1, 48:56] = -1.0                                    # full-scale plateaus: exact bin-edge style inputs
    cond_ref, mi, ai = orc.wsr_cond(c, mu_w, ang_w, return_idx=True)
    ct = T(c, dev)
    cond = npy(engine.wsr_cond(ct, T(mu_w, dev), T(ang_w, dev)))
    assert torch.equal(ct, T(c, dev))                          # the C ABI does not clip in place (the module does)
    assert cond.shape == cond_ref.shape == (B, 3659, L // 8)
    assert np.abs(cond[:, 3200:3209] - cond_ref[:, 3200:3209]).max() < 2e-6 * max(1.0, float(np.abs(cond_ref[:, 3200:3209]).max()))
    # the embeddings are table rows: equal bit for bit wherever the quantiser decisions agree; a decision may differ from the
    # oracle's only when the pre-rounding value sits within float noise of a bin edge -- allow a handful, never a pattern
    diff = (cond != cond_ref)
    diff[:, 3200:3209] = False
    bad_frames = np.unique(np.argwhere(diff)[:, [0, 2]], axis=0) if diff.any() else np.zeros((0, 2), int)
    assert len(bad_frames) <= max(1, (B * L // 8) // 500), "too many frames with a different quantiser decision: %d" % len(bad_frames)


@pytest.mark.parametrize("B,L", [(1, 64), (3, 1208)])
def test_wsr_cond_backward_vs_oracle(dev, B, L):
    from constant_memory_waveglow_amd import engine
    c = fill.uniform("wsrb/c%d_%d" % (B, L), (B, L), -1.1, 1.1)
    dcond = fill.normal("wsrb/g%d_%d" % (B, L), (B, 3659, L // 8))
    dmu_ref, dang_ref = orc.wsr_cond_backward(c, dcond, double=True)
    dmu, dang = engine.wsr_cond_backward(T(c, dev), T(dcond, dev))
    assert relmax(npy(dmu), dmu_ref) < 1e-5
    assert relmax(npy(dang), dang_ref) < 1e-5


@pytest.mark.parametrize("name", ["wsr", "wsr3"])
def test_wsrglow_model_vs_reference_golden(dev, golden_dir, name):
    """"wsr": WSRGlow(upsample_rate=2) (configs/wsrglow_vctk_2x.json); "wsr3": rate 3 (wsrglow_vctk_3x.json: n_group = hop = 24,
    1x1 convs of 24 / 22 / 20 channels)."""
    cfg = fill.CONFIGS[name]
    B, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    P.update(fill.wsr_tables(name + "/"))
    m = cm.WSRGlow(upsample_rate=fill.WSR_RATE[name], memory_efficient=True, bias=False, **fill.WSR_KW)
    sd = {k: torch.from_numpy(v) for k, v in P.items()}
    sd["window"] = torch.hann_window(16)
    m.load_state_dict(sd)
    m = m.to(dev)
    audio, c = fill.wsr_inputs(name, B, N, fill.WSR_RATE[name])
    gold = np.load(os.path.join(golden_dir, "model_%s.npz" % name))
    ct = T(c, dev)
    z, logdet = m(T(audio, dev), ct)
    assert float(ct.abs().max()) <= 1.0                        # clipped in place, as upstream (wsrglow.py:38)
    loss = cm.WaveGlowLoss(1.0)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - gold["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), gold["logdet"], N)
    assert abs(float(loss) - float(gold["loss"])) < LOSS_ATOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        g = npy(named[n].grad)
        nh = min(g.size, gold["grad_head"].shape[1])
        assert np.abs(g.ravel()[:nh] - gold["grad_head"][i][:nh]).max() / max(float(gold["grad_max"][i]), 1e-30) < GRAD_RTOL, n
        gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        assert abs(gn - float(gold["grad_norm"][i])) <= 1e-4 * float(gold["grad_norm"][i]) + 1e-12, n
    for n, _ in fill.WSR_TABLES:
        assert relmax(npy(named[n].grad), gold["grad::" + n]) < GRAD_RTOL, n
    with torch.no_grad():
        x, ld = m.reverse(T(gold["z"], dev), T(c, dev))
    assert np.abs(npy(x) - gold["x_inv"]).max() < Z_ATOL
    assert np.abs(npy(x) - audio).max() < Z_ATOL
    assert logdet_close(npy(ld), gold["logdet_inv"], N)


def test_wsrglow_timed_workload_vs_reference_golden(dev, golden_dir, precision, monkeypatch):
    """`bench.py --model wsrglow` at its own size -- WSRGlow(upsample_rate=2) as configs/wsrglow_vctk_2x.json ships it, 229.7 M parameters,
    batch 12 x 8192 -- through the TIMED path (FlowTrainer.step = wg_train_step) against one training step of the reference itself on the
    CPU (model_wsr_full.npz, a summary: loss, logdet, both ends and the per-item norm of z, norm / head of every gradient, the table
    gradients in full, both index arrays of the conditioning).  This is the shape at which the gate conv is cut along K (the counter
    must show it).

    49 152 low-rate samples and 55 296 STFT bins are quantised on the way in; a decision whose pre-rounding value sits within float noise of
    a bin edge may fall the other way on the GPU (here: ONE phase bin of item 0, frame 438).  The test finds such decisions from the
    recorded indices, allows two, and hands the flow the reference's table rows at those frames -- what is compared is then the same function
    -- skipping only the table-gradient rows the differing decisions themselves route to."""
    if precision != "bf16x3p":
        pytest.skip("the timed workload runs in the default arithmetic")
    from constant_memory_waveglow_amd import _lib, engine
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    name = "wsr_full"
    cfg = fill.CONFIGS[name]
    B, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    P.update(fill.wsr_tables(name + "/"))
    m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False)
    sd = {k: torch.from_numpy(v) for k, v in P.items()}
    sd["window"] = torch.hann_window(16)
    m.load_state_dict(sd)
    mu_t, ang_t = P["mu_enc.1.weight"], P["angle_embed.embed.weight"]
    del sd, P
    m = m.to(dev)
    audio, c = fill.wsr_inputs(name, B, N, 2)
    gold = np.load(os.path.join(golden_dir, "model_%s.npz" % name))
    # ---- the quantiser decisions
    Fr = c.shape[1] // 8
    real_cond = engine.wsr_cond
    cond = npy(real_cond(T(np.clip(c, -1.0, 1.0), dev), m.mu_enc[1].weight.detach(), m.angle_embed.embed.weight.detach()))
    ref_mu = mu_t[gold["mu_idx"].astype(np.int64)].reshape(B, Fr, 3200).transpose(0, 2, 1)
    ref_ang = ang_t[gold["ang_idx"].astype(np.int64)].transpose(0, 1, 3, 2).reshape(B, 450, Fr)
    bad_mu, bad_ang = np.argwhere((cond[:, :3200] != ref_mu).any(1)), np.argwhere((cond[:, 3209:] != ref_ang).any(1))
    assert len(bad_mu) + len(bad_ang) <= 2, (bad_mu.tolist(), bad_ang.tolist())
    # only a genuine tie is forgiven: at every differing decision the value the GPU truncates (wg_wsr_cond_pre: the functions
    # wsr_cond_kernel itself truncates) must sit within 1e-4 of an integer -- a bin edge -- and the two decisions must be the two sides of it
    mu_pre, ang_pre = (npy(t) for t in engine.wsr_cond_pre(T(np.clip(c, -1.0, 1.0), dev)))
    for b, f in bad_mu:
        for s_ in range(8):
            ref_i, pre = int(gold["mu_idx"][b, 8 * f + s_]), float(mu_pre[b, 8 * f + s_])
            if int(np.floor(pre)) != ref_i:
                assert abs(pre - round(pre)) < 1e-4 and abs(int(np.floor(pre)) - ref_i) == 1, (b, f, s_, pre, ref_i)
    for b, f in bad_ang:
        for k in range(9):
            ref_i, pre = int(gold["ang_idx"][b, k, f]), float(ang_pre[b, k, f])
            if int(pre) != ref_i:
                assert abs(pre - round(pre)) < 1e-4 and abs(int(pre) - ref_i) == 1, (b, k, f, pre, ref_i)
    # and the debug output is what the kernel used: truncating it reproduces the kernel's own choice of table row everywhere
    mu_rows = np.floor(mu_pre).astype(np.int64).reshape(B, Fr, 8)
    assert np.array_equal(cond[:, :3200].reshape(B, 8, 400, Fr), mu_t[mu_rows].transpose(0, 2, 3, 1))
    assert np.array_equal(cond[:, 3209:].reshape(B, 9, 50, Fr), ang_t[ang_pre.astype(np.int64)].transpose(0, 1, 3, 2))
    skip_rows = {"mu_enc.1.weight": set(), "angle_embed.embed.weight": set()}
    for b, f in bad_mu:
        for s_ in range(8):
            ours = cond[b, s_ * 400:(s_ + 1) * 400, f]
            skip_rows["mu_enc.1.weight"] |= {int(gold["mu_idx"][b, 8 * f + s_]), int(np.argmin(np.abs(mu_t - ours).sum(1)))}
    for b, f in bad_ang:
        for k in range(9):
            ours = cond[b, 3209 + k * 50:3209 + (k + 1) * 50, f]
            skip_rows["angle_embed.embed.weight"] |= {int(gold["ang_idx"][b, k, f]), int(np.argmin(np.abs(ang_t - ours).sum(1)))}

    def cond_with_reference_decisions(cc, mu_w, ang_w):
        h = real_cond(cc, mu_w, ang_w)
        for b, f in bad_mu:
            h[b, :3200, f] = T(np.ascontiguousarray(ref_mu[b, :, f]), dev)
        for b, f in bad_ang:
            h[b, 3209:, f] = T(np.ascontiguousarray(ref_ang[b, :, f]), dev)
        return h
    monkeypatch.setattr(engine, "wsr_cond", cond_with_reference_decisions)
    # ---- the timed step
    tr = FlowTrainer(m, 1.0)
    before = _lib.lib().wg_stat_gate_split_launches()
    loss, z, logdet = tr.step(T(audio, dev), T(c, dev))
    torch.cuda.synchronize()
    assert _lib.lib().wg_stat_gate_split_launches() - before == (2 * len(m.WNs) - 1) * 8
    zz = npy(z)
    assert np.abs(zz[:, :256] - gold["z_head"]).max() < Z_ATOL and np.abs(zz[:, -256:] - gold["z_tail"]).max() < Z_ATOL
    zn = np.sqrt((zz.astype(np.float64) ** 2).sum(1))
    assert np.all(np.abs(zn - gold["z_item_norm"]) <= 1e-5 * gold["z_item_norm"])
    assert logdet_close(npy(logdet), gold["logdet"], N)
    assert abs(float(loss) - float(gold["loss"])) < LOSS_ATOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        g = npy(named[n].grad)
        nh = min(g.size, gold["grad_head"].shape[1])
        assert np.abs(g.ravel()[:nh] - gold["grad_head"][i][:nh]).max() / max(float(gold["grad_max"][i]), 1e-30) < GRAD_RTOL, n
        gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        assert abs(gn - float(gold["grad_norm"][i])) <= 1e-4 * float(gold["grad_norm"][i]) + 1e-12, n
    for n, _ in fill.WSR_TABLES:
        keep = np.array([r for r in range(gold["grad::" + n].shape[0]) if r not in skip_rows[n]])
        assert relmax(npy(named[n].grad)[keep], gold["grad::" + n][keep]) < GRAD_RTOL, n


def test_wsrglow_full_width_vs_oracle(dev, precision):
    """The shipped WSRGlow width (WN 256 channels x 8 layers, 229.7 M parameters, V = 3659 -> 4096 per flow) on one short segment
    against the oracle: the conditioning GEMM segment with K = 3659 is the shape the small fixtures do not reach."""
    if precision != "bf16x3p":
        pytest.skip("full-width case runs in the default arithmetic only (CPU oracle time)")
    name = "wsr_full"
    cfg = dict(fill.CONFIGS["wsr"], dilation_channels=256, residual_channels=256, skip_channels=256, depth=8)
    B, N = 1, 1024
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    tabs = fill.wsr_tables(name + "/")
    audio, c = fill.wsr_inputs(name, B, N)
    cond = orc.wsr_cond(c, tabs["mu_enc.1.weight"], tabs["angle_embed.embed.weight"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, cond, 1.0, need_dh=True)
    dmu_ref, dang_ref = orc.wsr_cond_backward(c, ref["dh"])
    m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False)
    sd = {k: torch.from_numpy(v) for k, v in P.items()}
    sd.update({k: torch.from_numpy(v) for k, v in tabs.items()})
    sd["window"] = torch.hann_window(16)
    m.load_state_dict(sd)
    m = m.to(dev)
    z, logdet = m(T(audio, dev), T(c, dev))
    loss = cm.WaveGlowLoss(1.0)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    assert abs(float(loss) - ref["loss"]) < LOSS_ATOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n
    assert relmax(npy(named["mu_enc.1.weight"].grad), dmu_ref) < GRAD_RTOL
    assert relmax(npy(named["angle_embed.embed.weight"].grad), dang_ref) < GRAD_RTOL


# ---- trainer parity: Adam on the flat buffers vs torch.optim.Adam, the optimizer the reference instantiates ----

@pytest.mark.parametrize("wd,betas", [(0.0, (0.9, 0.999)), (0.01, (0.9, 0.98))])
def test_flat_adam_matches_torch_adam(dev, precision, wd, betas):
    if precision != "bf16x3p":
        pytest.skip("optimizer arithmetic does not depend on the contraction mode")
    from constant_memory_waveglow_amd.parallel import FlowTrainer, FlatAdam
    m, cfg, specs, P = build("micro", dev)
    B, N, F = fill.SHAPES["micro"]
    audio, h = fill.inputs("micro", B, N, F, cfg["n_mels"])
    tr = FlowTrainer(m, fill.SIGMA)
    names = [n for n, p in m.named_parameters()]
    ref_params = [torch.nn.Parameter(torch.from_numpy(P[n]).clone()) for n in names]
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3, betas=betas, eps=1e-8, weight_decay=wd)
    opt = FlatAdam(tr, lr=1e-3, betas=betas, eps=1e-8, weight_decay=wd)
    named = dict(m.named_parameters())
    assert all(named[n].data_ptr() >= opt.flat.data_ptr() for n in names)          # parameters live in the flat buffer now
    losses = []
    for step in range(3):
        loss, _, _ = tr.step(T(audio, dev), T(h, dev))
        losses.append(float(loss))
        for n, rp in zip(names, ref_params):
            rp.grad = named[n].grad.detach().cpu().clone()                        # same gradients: isolates the optimizer
        ref_opt.step()
        for n, rp in zip(names, ref_params):
            got, want = npy(named[n]), rp.detach().numpy()
            assert np.abs(got - want).max() <= 2e-6 * max(1.0, float(np.abs(want).max())), (step, n)
    assert losses[2] < losses[0]                                                   # and the step actually trains
    # optimizer checkpoints interchange with torch.optim.Adam
    sd = opt.state_dict()
    ref_sd = ref_opt.state_dict()
    assert set(sd["state"].keys()) == set(ref_sd["state"].keys())
    order = {id(p): i for i, p in enumerate(t for t in tr.table if t is not None)}
    for i, n in enumerate(names):
        j = order[id(named[n])]
        assert relmax(sd["state"][j]["exp_avg"].cpu().numpy(), ref_sd["state"][i]["exp_avg"].numpy()) < 1e-5, n
        assert relmax(sd["state"][j]["exp_avg_sq"].cpu().numpy(), ref_sd["state"][i]["exp_avg_sq"].numpy()) < 1e-5, n
    opt2_trainer = FlowTrainer(m, fill.SIGMA)
    opt2 = FlatAdam(opt2_trainer, lr=5e-4)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 3 and opt2.lr == 1e-3 and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)


# ---- shape sweep: configurations the fixtures do not hit, HIP engine vs oracle --------------------------------------------

SWEEP = [
    # flows, n_group, n_early_every, n_early_size, hop, n_mels, channels (dil, res, skip), depth, radix, B, frames
    dict(flows=2, n_group=4, n_early_every=1, n_early_size=2, hop_size=8, n_mels=8, ch=(32, 32, 32), depth=1, radix=3, B=1, F=5),
    dict(flows=3, n_group=6, n_early_every=2, n_early_size=2, hop_size=24, n_mels=33, ch=(64, 32, 96), depth=2, radix=3, B=3, F=7),
    dict(flows=5, n_group=12, n_early_every=2, n_early_size=4, hop_size=48, n_mels=17, ch=(32, 64, 32), depth=4, radix=3, B=2, F=9),
    dict(flows=2, n_group=32, n_early_every=4, n_early_size=2, hop_size=32, n_mels=40, ch=(96, 96, 64), depth=3, radix=3, B=1, F=300),
    dict(flows=4, n_group=8, n_early_every=3, n_early_size=2, hop_size=16, n_mels=80, ch=(32, 32, 32), depth=9, radix=3, B=5, F=70),
    dict(flows=2, n_group=8, n_early_every=4, n_early_size=2, hop_size=64, n_mels=20, ch=(64, 64, 64), depth=2, radix=1, B=2, F=4),
    # 37 time tiles x 7 items: 518 tiles (128x128 gate conv, 128x64 elsewhere) for 512 persistent workgroups -> some walk two tiles
    dict(flows=1, n_group=8, n_early_every=4, n_early_size=2, hop_size=64, n_mels=20, ch=(128, 128, 128), depth=2, radix=3, B=7, F=592),
    # the architecture behind the one speed the reference publishes (configs/musicnet_config.json:7-20, README.md:64-67): 18 flows, early
    # outputs every 6, WN depth 4, hop 512 (an upsampler of stride 64, 129 taps); bench.py's `inverse_khz_musicnet_*` leg times it
    dict(flows=18, n_group=8, n_early_every=6, n_early_size=2, hop_size=512, n_mels=80, ch=(256, 256, 256), depth=4, radix=3, B=1, F=6),
]


@pytest.mark.parametrize("case", range(len(SWEEP)))
def test_shape_sweep_vs_oracle(dev, case):
    """Odd channel mixes (dilation != residual != skip), depth 1 and 9 (dilation 256 > T), n_group 4..32, early outputs every flow,
    radix 1, a time axis shorter than one tile and one that is not a multiple of it, batch 1..7, a launch with a few more tiles
    than persistent workgroups, and the reference's musicnet architecture (18 flows of depth 4, hop 512)."""
    c = SWEEP[case]
    cfg = dict(flows=c["flows"], n_group=c["n_group"], n_early_every=c["n_early_every"], n_early_size=c["n_early_size"],
               hop_size=c["hop_size"], n_mels=c["n_mels"], dilation_channels=c["ch"][0], residual_channels=c["ch"][1],
               skip_channels=c["ch"][2], depth=c["depth"], radix=c["radix"])
    tag = "sweep%d" % case
    B, F = c["B"], c["F"]
    N = F * c["hop_size"] - (c["hop_size"] // 2 if case % 2 else 0)         # odd cases: audio shorter than the mel covers
    N -= N % c["n_group"]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, tag + "/")
    audio, h = fill.inputs(tag, B, N, F, cfg["n_mels"])
    ref = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True)
    m = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    assert m.z_split_sizes and sum(m.z_split_sizes) == cfg["n_group"]
    ht = T(h, dev).requires_grad_(True)
    z, logdet = m(T(audio, dev), ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    # the loss is a sum over B*N terms: the oracle's own fp32 accumulation is off by ~1e-6 at 265 000 terms (case 6), so the loss is
    # checked against the float64 evaluation of the formula on the oracle's z / logdet (model/loss.py:10-15)
    z64, ld64 = ref["z"].astype(np.float64), ref["logdet"].astype(np.float64)
    loss64 = float(np.mean(0.5 * (z64 * z64).sum(1) / fill.SIGMA ** 2 - ld64) / N)
    assert abs(loss64 - ref["loss"]) < 3 * LOSS_ATOL
    assert abs(float(loss) - loss64) < LOSS_ATOL
    assert relmax(npy(ht.grad), ref["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        g = npy(named[n].grad)
        if n.endswith("weight_v") and g[0].size == 1:
            # fan-in 1 (start conv of a 2-channel flow): w = g * sign(v), so dL/dv is exactly zero and both sides hold rounding noise
            gg = np.abs(npy(named[n[:-1] + "g"].grad)).max()
            assert np.abs(g).max() < 1e-6 * gg and np.abs(ref["grads"][i]).max() < 1e-6 * gg, n
            continue
        assert relmax(g, ref["grads"][i]) < GRAD_RTOL, n
    with torch.no_grad():
        x, ld = m.reverse(z.detach(), ht.detach())
    assert np.abs(npy(x) - audio).max() < Z_ATOL
    assert logdet_close(-npy(ld), ref["logdet"], N)


# ---- WaveFlow ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["wf8", "wf64", "wf8c", "wf64c", "wf8b", "wf64b"])
def test_waveflow_model_vs_reference_golden(dev, golden_dir, precision, name):
    """WaveFlow forward + NLL + backward + row-by-row inverse against the reference's own run (model_wf*.npz) and the oracle.
    "wf8c" / "wf64c": use_conv1x1=True (an invertible 1x1 conv over the height axis instead of the flip, waveflow.py:203-206);
    "wf8b" / "wf64b": WN2D(bias=True) (waveflow.py:77): the ones segment behind nine taps + conditioning, every bias gradient."""
    from oracle import wf_oracle as wfo
    cfg = fill.WF_CONFIGS[name]
    B, N, F = fill.WF_SHAPES[name]
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    gold = np.load(os.path.join(golden_dir, "model_%s.npz" % name))
    ref = wfo.train_step(wfo.make_config(**cfg), fill.table(specs, P), audio, mel, fill.SIGMA, need_dmel=True)
    conv = bool(cfg.get("use_conv1x1"))
    m = cm.WaveFlow(memory_efficient=False, **dict({"use_conv1x1": False, "bias": False}, **cfg))
    if cfg.get("bias"):                                                               # (the specs list the biases behind end.weight: table order)
        assert sorted(n for n, _ in m.named_parameters()) == sorted(n for n, _, _ in specs)
    else:
        assert [n for n, _ in m.named_parameters()] == [n for n, _, _ in specs]      # invconv1x1.* after WNs.*, as upstream
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    ht = T(mel, dev).requires_grad_(True)
    z, logdet = m(T(audio, dev), ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    for want in (ref, gold):
        assert np.abs(npy(z) - want["z"]).max() < Z_ATOL
        assert logdet_close(npy(logdet), want["logdet"], N)
        assert abs(float(loss) - float(want["loss"])) < LOSS_ATOL
        assert relmax(npy(ht.grad), want["dmel"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        g = npy(named[n].grad)
        if n.endswith("start.weight_v"):
            # Conv2d(1, C, 1) under weight norm: the exact gradient w.r.t. v is zero (w = g * sign(v)); rounding noise on every side
            assert np.abs(g).max() < 1e-5 * np.abs(npy(named[n[:-1] + "g"].grad)).max(), n
            continue
        assert relmax(g, ref["grads"][i]) < GRAD_RTOL, n
        nh = min(g.size, gold["grad_head"].shape[1])
        assert np.abs(g.ravel()[:nh] - gold["grad_head"][i][:nh]).max() / max(float(gold["grad_max"][i]), 1e-30) < GRAD_RTOL, n
        if "grad::" + n in gold:
            assert relmax(g, gold["grad::" + n]) < GRAD_RTOL, n
    with torch.no_grad():
        x, ld = m.reverse(T(gold["z"], dev), ht.detach())
        y_up = npy(m._upsample_h(ht.detach()))
    assert conv == hasattr(m, "invconv1x1")
    assert np.abs(npy(x) - gold["x_inv"]).max() < Z_ATOL
    assert np.abs(npy(x) - audio).max() < Z_ATOL
    assert logdet_close(npy(ld), gold["logdet_inv"], N)
    assert y_up.shape == gold["y_up"].shape and np.abs(y_up - gold["y_up"]).max() < 2e-6     # WaveFlow._upsample_h (waveflow.py:255-257)


@pytest.mark.parametrize("name", ["wf8", "wf8c"])
def test_waveflow_reverse_mode_vs_reference_golden(dev, golden_dir, precision, name):
    """WaveFlow(reverse_mode=True) against the reference's own run (make_golden.waveflow_rm_fixture): `forward` is the row loop, `reverse` and
    `infer` the parallel map (differentiable), the 1x1 convs swap direction with the model (base.py:20-28, efficient_modules.py:30-56)."""
    cfg = fill.WF_CONFIGS[name]
    B, N, F = fill.WF_SHAPES[name]
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    gold = np.load(os.path.join(golden_dir, "model_%s_rm.npz" % name))
    m = cm.WaveFlow(memory_efficient=False, reverse_mode=True, bias=False, **dict({"use_conv1x1": False}, **cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    with torch.no_grad():
        z, ld = m(T(audio, dev), T(mel, dev))
    assert np.abs(npy(z) - gold["z"]).max() < Z_ATOL
    assert logdet_close(npy(ld), gold["logdet"], N)
    ht = T(mel, dev).requires_grad_(True)
    x, ldx = m.reverse(T(gold["z"], dev), ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(x, ldx)
    loss.backward()
    assert np.abs(npy(x) - gold["x_rev"]).max() < Z_ATOL and np.abs(npy(x) - audio).max() < Z_ATOL
    assert logdet_close(npy(ldx), gold["logdet_rev"], N)
    assert abs(float(loss) - float(gold["loss"])) < LOSS_ATOL
    assert relmax(npy(ht.grad), gold["dmel"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        if n.endswith("start.weight_v"):
            continue
        g = npy(named[n].grad).astype(np.float64)
        assert abs(np.sqrt((g ** 2).sum()) - float(gold["grad_norm"][i])) <= GRAD_RTOL * float(gold["grad_norm"][i]) + 1e-12, n
        if "grad::" + n in gold:
            assert relmax(g, gold["grad::" + n]) < GRAD_RTOL, n
    torch.manual_seed(3)
    y = m.infer(T(mel, dev)[0], 0.7)                              # infer = the parallel map on a fresh latent
    assert y.shape == (mel.shape[2] * 256,) and bool(torch.isfinite(y).all())


def test_waveflow_two_forwards_before_backward(dev, precision):
    """Two forwards of the same shape, then backward through BOTH (two losses / gradient accumulation): each autograd node owns
    the tape of its own forward, so the first backward recomputes from the first call's flow inputs, not the second's."""
    name = "wf8"
    cfg = fill.WF_CONFIGS[name]
    B, N, F = fill.WF_SHAPES[name]
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    audio2 = np.ascontiguousarray(audio[::-1] * 0.5)
    mel2 = np.ascontiguousarray(mel[::-1] + 0.25)
    m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    crit = cm.WaveGlowLoss(fill.SIGMA)
    want = []
    for a, me in ((audio, mel), (audio2, mel2)):
        m.zero_grad(set_to_none=True)
        z, ld = m(T(a, dev), T(me, dev))
        crit(z, ld).backward()
        want.append([p.grad.clone() for p in m.parameters()])
    m.zero_grad(set_to_none=True)
    za, lda = m(T(audio, dev), T(mel, dev))
    zb, ldb = m(T(audio2, dev), T(mel2, dev))                 # same shape: the old engine-owned tape was overwritten here
    crit(za, lda).backward()
    first = [p.grad.clone() for p in m.parameters()]
    for g, w in zip(first, want[0]):
        assert torch.equal(g, w)
    crit(zb, ldb).backward()                                   # accumulates
    for p, w0, w1 in zip(m.parameters(), want[0], want[1]):
        assert float((p.grad - (w0 + w1)).abs().max()) <= 1e-6 * float((w0 + w1).abs().max()) + 1e-12
    with torch.no_grad():                                      # no gradient wanted: no tape is kept
        _, _, tape = m._engine.forward([None if t is None else t.detach() for t in m.param_table()], T(audio, dev), T(mel, dev), False)
    assert tape is None


def test_waveflow_timed_workload_vs_reference_golden(dev, golden_dir, precision):
    """`bench.py --model waveflow` at its own size -- configs/waveflow_LJ_speech.json: 8 flows, 64 rows, 64 channels, batch 12 x 16000 --
    against one training step of the reference itself on the CPU (model_wf_full.npz, a summary: both ends and the per-item norm of z,
    logdet, loss, norm / head of every gradient, d mel in full)."""
    if precision != "bf16x3p":
        pytest.skip("the timed workload runs in the default arithmetic")
    cfg = dict(flows=8, n_group=64, n_mels=80, dilation_channels=64, residual_channels=64, skip_channels=64)
    B, N, F = 12, 16000, 63
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, "wf_full/")
    m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    audio, mel = fill.waveflow_inputs("wf_full", B, N, F, 80)
    gold = np.load(os.path.join(golden_dir, "model_wf_full.npz"))
    ht = T(mel, dev).requires_grad_(True)
    z, logdet = m(T(audio, dev), ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    _check_summary(gold, m, specs, z, logdet, loss, N)
    assert relmax(npy(ht.grad), gold["dmel"]) < GRAD_RTOL


def test_waveflow_timed_workload_vs_float64_oracle(dev, precision):
    """`bench.py --model waveflow` at its own size (configs/waveflow_LJ_speech.json: 8 flows, 64 rows, 64 channels, batch 12 x 16000) against
    the float64 torch-CPU oracle IN FULL -- every element of z, logdet, loss, d loss / d mel and every one of the 299 gradients -- where the
    reference's own step at this size is kept as a summary only (test_waveflow_timed_workload_vs_reference_golden).  The oracle
    (oracle/torch_cpu.waveflow_train_step, pinned to the reference's goldens by tests/test_oracle_golden.py) runs one worker process per
    share of the batch.  Bars: z 1e-4, logdet rtol 1e-4, loss 1e-6, every gradient within 1e-4 of its tensor's max."""
    if precision != "bf16x3p":
        pytest.skip("the timed workload runs in the default arithmetic (CPU oracle time)")
    from oracle import torch_cpu
    cfg = dict(flows=8, n_group=64, n_mels=80, dilation_channels=64, residual_channels=64, skip_channels=64)
    B, N, F = 12, 16000, 63
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, "wf_full/")
    audio, mel = fill.waveflow_inputs("wf_full", B, N, F, 80)
    n_allowed, quota, firsts = torch_cpu.host_cpu_budget()      # (what this process may really use: a container may grant a fraction of what it shows)
    cores = max(1, int(quota) if quota else len(firsts))
    workers = max(1, min(B, cores // 2))
    ref = torch_cpu.train_step_parallel(dict(cfg, model="waveflow"), fill.table(specs, P), audio, mel, fill.SIGMA, workers=workers,
                                        threads=max(1, min(4, cores // workers)), need_dh=True, double=True)
    m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    ht = T(mel, dev).requires_grad_(True)
    z, logdet = m(T(audio, dev), ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    assert abs(float(loss) - ref["loss"]) < LOSS_ATOL
    assert relmax(npy(ht.grad).astype(np.float64), ref["dh"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    worst = 0.0
    for i, (n, _, _) in enumerate(specs):
        g = npy(named[n].grad).astype(np.float64)
        if n.endswith("start.weight_v"):
            # Conv2d(1, C, 1) under weight norm: w = g sign(v), the exact gradient w.r.t. v is zero; both sides hold rounding noise
            assert np.abs(g).max() < 1e-5 * np.abs(npy(named[n[:-1] + "g"].grad)).max(), n
            continue
        e = relmax(g, ref["grads"][i])
        worst = max(worst, e)
        assert e < GRAD_RTOL, n
    print("WaveFlow 12 x 16000 vs float64 oracle: |dz| %.2e, worst gradient %.2e of its tensor's max (%d oracle workers)"
          % (float(np.abs(npy(z) - ref["z"]).max()), worst, workers))


def test_waveflow_full_size_properties(dev, precision):
    """BASELINE.json configs[3] at its full size (configs/waveflow_LJ_speech.json: 8 flows, 64 rows, 64 channels, batch 12 x 16000):
    size-independent properties instead of an oracle run."""
    torch.manual_seed(0)
    m = cm.WaveFlow(flows=8, n_group=64, n_mels=80, use_conv1x1=False, memory_efficient=False, dilation_channels=64,
                    residual_channels=64, skip_channels=64, bias=False)
    with torch.no_grad():
        for wn in m.WNs:
            wn.end.weight.normal_(0.0, 0.02)
    m = m.to(dev)
    B, N, F = 12, 16000, 63
    x = T(fill.uniform("wffull/x", (B, N), -1.0, 1.0), dev)
    h = T(fill.normal("wffull/h", (B, 80, F)), dev)
    crit = cm.WaveGlowLoss(1.0)
    z, logdet = m(x, h)
    loss = crit(z, logdet)
    loss.backward()
    assert bool(torch.isfinite(z).all()) and bool(torch.isfinite(logdet).all()) and bool(torch.isfinite(loss))
    g_full = {n: p.grad.clone() for n, p in m.named_parameters()}
    assert all(bool(torch.isfinite(g).all()) for g in g_full.values())
    # batch items are independent units
    with torch.no_grad():
        z1, ld1 = m(x[5:6].clone(), h[5:6])
    assert float((z1 - z[5:6]).abs().max()) < 1e-5
    assert abs(float(ld1[0] - logdet[5])) < 1e-4 * abs(float(logdet[5])) + 1e-3
    # forward o reverse = id on two items (the row-by-row inverse is the slow direction) and logdet_fwd = -logdet_rev
    with torch.no_grad():
        xr, ldr = m.reverse(z[:2].detach().clone(), h[:2])
    assert float((xr - x[:2]).abs().max()) < Z_ATOL
    assert float((logdet[:2].detach() + ldr).abs().max()) < 1e-4 * float(logdet[:2].abs().max()) + 1e-2
    # linearity of the gradient in the batch (DP semantics)
    m.zero_grad()
    for sl in (slice(0, 6), slice(6, 12)):
        zz, ll = m(x[sl].clone(), h[sl])
        (0.5 * crit(zz, ll)).backward()
    for n, p in m.named_parameters():
        if n.endswith("start.weight_v"):
            continue                                           # exact gradient zero: rounding noise on both sides
        a, b = p.grad, g_full[n]
        assert float((a - b).abs().max()) <= GRAD_RTOL * float(b.abs().max()) + 1e-12, n


def test_wsrglow_full_size_properties(dev, precision):
    """BASELINE.json configs[4] at its full size (configs/wsrglow_vctk_2x.json: 229.7 M parameters, batch 12 x 8192)."""
    if precision != "bf16x3p":
        pytest.skip("full-size case runs in the default arithmetic only")
    torch.manual_seed(0)
    m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False)
    with torch.no_grad():
        for blk in m.WNs:
            blk.F.end.weight.normal_(0.0, 0.02)
    m = m.to(dev)
    B, N = 12, 8192
    x = T(fill.uniform("wsrfull/x", (B, N), -1.0, 1.0), dev)
    c = T(fill.uniform("wsrfull/c", (B, N // 2), -0.95, 0.95), dev)
    crit = cm.WaveGlowLoss(1.0)
    z, logdet = m(x, c.clone())
    loss = crit(z, logdet)
    loss.backward()
    assert bool(torch.isfinite(z).all()) and bool(torch.isfinite(logdet).all()) and bool(torch.isfinite(loss))
    g_full = {n: p.grad.clone() for n, p in m.named_parameters()}
    assert all(bool(torch.isfinite(g).all()) for g in g_full.values())
    with torch.no_grad():
        xr, ldr = m.reverse(z.detach(), c.clone())
        z1, ld1 = m(x[5:6].clone(), c[5:6].clone())
    assert float((xr - x).abs().max()) < Z_ATOL
    assert float((logdet.detach() + ldr).abs().max()) < 1e-4 * float(logdet.abs().max()) + 1e-2
    assert float((z1 - z[5:6]).abs().max()) < 1e-5
    assert abs(float(ld1[0] - logdet[5])) < 1e-4 * abs(float(logdet[5])) + 1e-3
    m.zero_grad()
    for sl in (slice(0, 6), slice(6, 12)):
        zz, ll = m(x[sl].clone(), c[sl].clone())
        (0.5 * crit(zz, ll)).backward()
    for n, p in m.named_parameters():
        a, b = p.grad, g_full[n]
        assert float((a - b).abs().max()) <= GRAD_RTOL * float(b.abs().max()) + 1e-12, n
    # the trainer's bucketed step on the same batch gives the same gradients
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    tr = FlowTrainer(m, 1.0)
    tr.step(x, c.clone())
    for n, p in m.named_parameters():
        if n in ("mu_enc.1.weight", "angle_embed.embed.weight"):       # LDS float atomics: equal to rounding, not bit for bit
            assert float((p.grad - g_full[n]).abs().max()) <= 1e-5 * float(g_full[n].abs().max()), n
        else:
            assert torch.equal(p.grad, g_full[n]), n


def test_wsrglow_gate_conv_cut_along_k_vs_uncut(dev, precision, monkeypatch):
    """WSRGlow's gate conv at the timed shape (M 512, K 4432 = 139 chunks, 12 x 512 columns = 32 column tiles x 2 row tiles) fills a
    quarter of the CUs, so run_convgemm cuts K in 4 (convgemm16g_kernel<8> + gate_finish16g_kernel, csrc/wg_gemm16g.h).  The cut changes
    only the fp32 summation order: a step must agree with the uncut kernels (WG_G192_SPLITK=0) to rounding, repeat bit for bit, and
    the counter must show the cut ran (every gate conv of the forward and of the recompute)."""
    if precision != "bf16x3p":
        pytest.skip("the LDS-DMA kernels exist in the S-plane mode only")
    from constant_memory_waveglow_amd import _lib
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    torch.manual_seed(0)
    m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False)
    with torch.no_grad():
        for blk in m.WNs:
            blk.F.end.weight.normal_(0.0, 0.02)
    m = m.to(dev)
    B, N = 12, 8192
    x = T(fill.uniform("wsrfull/x", (B, N), -1.0, 1.0), dev)
    c = T(fill.uniform("wsrfull/c", (B, N // 2), -0.95, 0.95), dev)
    tr = FlowTrainer(m, 1.0)
    res = {}
    for cut in ("1", "0"):
        monkeypatch.setenv("WG_G192_SPLITK", cut)
        cm._lib.lib().wg_reload_env()
        before = _lib.lib().wg_stat_gate_split_launches()
        runs = []
        for rep in range(2):
            loss, z, logdet = tr.step(x, c.clone())
            runs.append((loss.clone(), z.clone(), logdet.clone(), tr.fg.flat.clone()))
        torch.cuda.synchronize()
        n = _lib.lib().wg_stat_gate_split_launches() - before
        flows, layers = len(m.WNs), 8
        assert n == (2 * (2 * flows - 1) * layers if cut == "1" else 0), n
        for a, b in zip(runs[0][:3], runs[1][:3]):
            assert torch.equal(a, b)
        # (the two embedding tables' gradients go through LDS float atomics: equal to rounding; everything else repeats exactly)
        assert float((runs[0][3] - runs[1][3]).abs().max()) <= 1e-5 * float(runs[0][3].abs().max())
        res[cut] = runs[0]
    (l1, z1, d1, g1), (l0, z0, d0, g0) = res["1"], res["0"]
    assert abs(float(l1) - float(l0)) < 1e-6 and float((z1 - z0).abs().max()) < 2e-5
    assert float((d1 - d0).abs().max()) <= 1e-6 * float(d0.abs().max()) + 1e-3
    assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max())


def test_waveflow_shipped_width_vs_oracle(dev, precision):
    """The shipped WaveFlow width (8 flows, 64 rows, 80 mels, 64 channels: one 128-row tile, K = 9*64 + 96) on a short segment."""
    from oracle import wf_oracle as wfo
    name = "wf_full"
    cfg = dict(flows=8, n_group=64, n_mels=80, dilation_channels=64, residual_channels=64, skip_channels=64)
    B, N, F = 1, 64 * 10, 2
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    ref = wfo.train_step(wfo.make_config(**cfg), fill.table(specs, P), audio, mel, fill.SIGMA, need_dmel=True)
    m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    ht = T(mel, dev).requires_grad_(True)
    z, logdet = m(T(audio, dev), ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    assert abs(float(loss) - ref["loss"]) < LOSS_ATOL
    assert relmax(npy(ht.grad), ref["dmel"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        if n.endswith("start.weight_v"):
            continue
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n
    with torch.no_grad():
        x, _ = m.reverse(z.detach(), ht.detach())
    assert np.abs(npy(x) - audio).max() < Z_ATOL


@pytest.mark.parametrize("cname", ["wf8", "wf64", "wf64_short", "wf8b"])
def test_wn2d_alone_vs_reference_golden(dev, golden_dir, precision, cname):
    """WN2D.forward on its own (model/waveflow.py:128-135) through wg_wf_wn_apply against the reference's own WN2D (block_wn2d.npz):
    the input has fewer rows than n_group (WaveFlow passes x[:, :, :-1]; 20 of 64 in the short case), the rest of the planes is zero."""
    from make_golden import wn2d_inputs
    cfg, P, x, y = wn2d_inputs(cname)
    gold = np.load(os.path.join(golden_dir, "block_wn2d.npz"))
    m = cm.waveflow.WN2D(cfg["n_group"], cfg["n_mels"], dilation_channels=cfg["dilation_channels"], residual_channels=cfg["residual_channels"],
                         skip_channels=cfg["skip_channels"], bias=bool(cfg.get("bias")), zero_init=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    with torch.no_grad():
        ls, t = m(T(x, dev), T(y, dev))
    assert ls.shape == t.shape == x.shape
    assert np.abs(npy(ls) - gold[cname + "/log_s"]).max() < Z_ATOL and np.abs(npy(t) - gold[cname + "/t"]).max() < Z_ATOL


@pytest.mark.parametrize("cname", ["wf8", "wf64_short", "wf8b"])
def test_wn2d_on_its_own_is_differentiable_vs_reference_golden(dev, golden_dir, precision, cname):
    """WN2D is an ordinary differentiable module upstream (model/waveflow.py:70-135): `log_s, t = wn2d(x, y)` followed by any loss gives
    gradients for x, y and every parameter.  Here the call is an autograd node (waveflow._WN2DFn) whose backward is wg_wf_wn_backward;
    checked against what the reference's own autograd gave (block_wn2d.npz: d x, d y in full, norm / head of every parameter gradient)
    and, in full, against autograd over oracle/torch_cpu.py's restatement."""
    from make_golden import wn2d_inputs, wn2d_seeds
    from oracle import torch_cpu
    cfg, P, x, y = wn2d_inputs(cname)
    gold = np.load(os.path.join(golden_dir, "block_wn2d.npz"))
    gls, gt = wn2d_seeds(cname, x.shape)
    m = cm.waveflow.WN2D(cfg["n_group"], cfg["n_mels"], dilation_channels=cfg["dilation_channels"], residual_channels=cfg["residual_channels"],
                         skip_channels=cfg["skip_channels"], bias=bool(cfg.get("bias")), zero_init=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ls, t = m(xt, yt)
    assert ls.requires_grad and t.requires_grad
    ((ls * T(gls, dev)).sum() + (t * t * T(gt, dev)).sum()).backward()
    assert np.abs(npy(ls) - gold[cname + "/log_s"]).max() < Z_ATOL and np.abs(npy(t) - gold[cname + "/t"]).max() < Z_ATOL
    assert relmax(npy(xt.grad), gold[cname + "/dx"]) < GRAD_RTOL and relmax(npy(yt.grad), gold[cname + "/dy"]) < GRAD_RTOL
    order = [n for n, _ in m.named_parameters()]
    tab = m.param_table()
    pt = [None if p is None else p.detach().cpu().clone().requires_grad_(True) for p in tab]
    xr, yr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(y).requires_grad_(True)
    ls_r, t_r = torch_cpu.wn2d_forward_t(pt, cfg["n_group"], xr, yr)
    ((ls_r * torch.from_numpy(gls)).sum() + (t_r * t_r * torch.from_numpy(gt)).sum()).backward()
    names = list(P)
    named = dict(m.named_parameters())
    by_id = {id(p): n for n, p in named.items()}
    for p, q in zip(tab, pt):
        n = by_id[id(p)]
        g = npy(p.grad)
        i = names.index(n)
        if n == "start.weight_v":                              # Conv2d(1, C, 1) under weight norm: exactly zero; rounding noise on every side
            assert np.abs(g).max() < 1e-5 * np.abs(npy(named["start.weight_g"].grad)).max(), n
            continue
        assert relmax(g, q.grad.numpy()) < GRAD_RTOL, n
        nh = min(g.size, gold[cname + "/grad_head"].shape[1])
        assert np.abs(g.ravel()[:nh] - gold[cname + "/grad_head"][i][:nh]).max() / max(float(gold[cname + "/grad_max"][i]), 1e-30) < GRAD_RTOL, n
    assert len(order) == len(names)


def test_waveflow_chip_filling_shape_vs_oracle(dev, precision):
    """The shipped WaveFlow width at a size that FILLS the chip -- batch 4 x 16000 samples = 256 plane rows x 250 columns -- against the
    C oracle: the launches the small fixtures never reach: 64-row tiles for the 64-row products (convgemm16q_kernel<.., M64>: the
    data-gradient conv, residual / skip products, gate backward), the gate conv on 128 x 256 column-group tiles (<.., CG2>), the grouped
    weight-gradient launch at K = 64 000 columns, wf_rowsum_s_kernel's four-way row split (configs/waveflow_LJ_speech.json is this
    network at batch 12)."""
    if precision != "bf16x3p":
        pytest.skip("the chip-filling WaveFlow case runs in the default arithmetic only (CPU oracle time)")
    from oracle import wf_oracle as wfo
    name = "wf_chip"
    cfg = dict(flows=8, n_group=64, n_mels=80, dilation_channels=64, residual_channels=64, skip_channels=64)
    B, N, F = 4, 16000, 63
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    ref = wfo.train_step(wfo.make_config(**cfg), fill.table(specs, P), audio, mel, fill.SIGMA, need_dmel=True)
    m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    ht = T(mel, dev).requires_grad_(True)
    z, logdet = m(T(audio, dev), ht)
    loss = cm.WaveGlowLoss(fill.SIGMA)(z, logdet)
    loss.backward()
    assert np.abs(npy(z) - ref["z"]).max() < Z_ATOL
    assert logdet_close(npy(logdet), ref["logdet"], N)
    assert abs(float(loss.detach()) - ref["loss"]) < LOSS_ATOL
    assert relmax(npy(ht.grad), ref["dmel"]) < GRAD_RTOL
    named = dict(m.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        if n.endswith("start.weight_v"):
            continue
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n


# ---- log-mel conditioner -----------------------------------------------------------------------------------

@pytest.mark.parametrize("B,N", [(1, 4096), (3, 16000), (2, 22016)])
def test_melspec_kernel_vs_oracle(dev, precision, B, N):
    """wg_melspec against the numpy restatement of MelSpec (reflection pad, 1024-point periodic-Hann STFT power, HTK mel filters to
    8 kHz, log(x + 1e-7)); the configuration is the one every WaveGlow / WaveFlow config ships (conditioner.args)."""
    if precision != "f32":
        pytest.skip("the conditioner does not depend on the contraction mode")
    from oracle import mel_oracle as mo
    x = fill.uniform("mel/x%d_%d" % (B, N), (B, N), -0.8, 0.8)
    x[0, : N // 4] *= 1e-3                                       # a quiet stretch: log-mel near its floor
    cond = cm.MelSpec(sr=22050, n_fft=1024, hop_length=256, f_max=8000, n_mels=80)
    got = npy(cond(T(x, dev)))
    want = mo.melspec(x, 22050, 1024, 256, 0.0, 8000.0, 80)
    assert got.shape == want.shape == (B, 80, N // 256 + 1)
    assert np.abs(got - want).max() < 1e-4                       # log domain, fp32 direct DFT vs float64 FFT
    with pytest.raises(cm.WgError):
        cm.MelSpec(sr=22050, n_fft=1024, hop_length=256, power=1.0)


@pytest.mark.parametrize("tag", list(fill.MEL_CASES))
def test_melspec_kernel_vs_reference_golden(dev, golden_dir, precision, tag):
    """wg_melspec against the reference's MelSpec class run over torch.stft (tests/golden/cond_melspec.npz): the power spectrogram
    before the filterbank -- reflection pad, periodic Hann, STFT, |.|^2: all torch's own code upstream -- and the log-mel output
    (whose filterbank is the published torchaudio formula on both sides)."""
    if precision != "f32":
        pytest.skip("the conditioner does not depend on the contraction mode")
    from constant_memory_waveglow_amd import engine
    G = np.load(os.path.join(golden_dir, "cond_melspec.npz"))
    kw = fill.MEL_KW
    x = fill.mel_input(tag)
    logmel, power = engine.melspec(T(x, dev), kw["sr"], kw["n_fft"], kw["hop_length"], 0.0, kw["f_max"], kw["n_mels"], return_power=True)
    want_p = G[tag + "/power"]
    assert tuple(power.shape) == want_p.shape
    assert np.abs(npy(power) - want_p).max() <= 2e-5 * float(want_p.max())       # fp32 direct DFT against torch's fp32 FFT
    assert np.abs(npy(logmel) - G[tag + "/logmel"]).max() < 2e-4                 # log domain, down to the 1e-7 floor
    cond = cm.MelSpec(sr=kw["sr"], n_fft=kw["n_fft"], hop_length=kw["hop_length"], f_max=kw["f_max"], n_mels=kw["n_mels"])
    assert torch.equal(cond(T(x, dev)), logmel)


# ---- race screen: the conv kernels' loader / compute protocol (hand-counted waits, one barrier per chunk) under repetition ---------

def test_repeated_steps_are_bitwise_identical(dev, precision):
    """Every kernel on the WaveGlow training path reduces in a fixed order, so the same inputs must give bit-identical outputs on
    every run; a race between the loader and compute waves (a staged register read before its wait, an LDS buffer overwritten too
    early) would show up as run-to-run differences long before it breaks a tolerance.  40 steps of C1 and 4 of the full-size model."""
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    m, cfg, specs, P = build("c1", dev)
    B, N, F = fill.SHAPES["c1"]
    audio, h = fill.inputs("c1", B, N, F, cfg["n_mels"])
    tr = FlowTrainer(m, fill.SIGMA)
    x, ht = T(audio, dev), T(h, dev)
    loss0, z0, ld0 = tr.step(x, ht)
    g0 = tr.fg.flat.clone()
    z0, ld0 = z0.clone(), ld0.clone()
    for _ in range(40):
        loss, z, ld = tr.step(x, ht)
        assert torch.equal(z, z0) and torch.equal(ld, ld0) and torch.equal(tr.fg.flat, g0)
    if precision != "bf16x3p":
        return
    import bench
    big = FlowTrainer(bench.build_model(dev), bench.SIGMA)
    g = torch.Generator(device=dev).manual_seed(7)
    xb = torch.rand(24, bench.SEG, device=dev, generator=g) * 2 - 1
    hb = torch.randn(24, 80, bench.FRAMES, device=dev, generator=g)
    _, zb0, _ = big.step(xb, hb)
    zb0, gb0 = zb0.clone(), big.fg.flat.clone()
    for _ in range(4):
        _, zb, _ = big.step(xb, hb)
        assert torch.equal(zb, zb0) and torch.equal(big.fg.flat, gb0)


@pytest.mark.parametrize("tag", list(fill.DECIMATE_CASES))
def test_stft_decimate_vs_reference_golden(dev, golden_dir, precision, tag):
    """wg_lowpass behind STFTDecimate against the reference's own output (cond_stftdecimate.npz) and the oracle."""
    if precision != "f32":
        pytest.skip("the conditioner does not depend on the contraction mode")
    from oracle import mel_oracle as mo
    B, Tn, r = fill.DECIMATE_CASES[tag]
    x = fill.uniform("decimate/" + tag, (B, Tn), -0.9, 0.9)
    gold = np.load(os.path.join(golden_dir, "cond_stftdecimate.npz"))[tag]
    got = npy(cm.STFTDecimate(r)(T(x, dev)))
    assert got.shape == gold.shape
    assert np.abs(got - gold).max() < 2e-5 and np.abs(got - mo.stft_decimate(x, r)).max() < 2e-5
    full = npy(cm.LowPass()(T(x, dev), 7))                      # ratio 1/1 keeps every bin incl. Nyquist: the identity up to rounding
    assert np.abs(full - x).max() < 2e-5


def test_half_inference_matches_fp32_engine(dev, precision):
    """`inference.py --half` (model.half(), cond.half(), inference.py:33-36): half storage in and out, fp32 arithmetic inside.
    The result equals the fp32 run on the half-rounded weights / inputs, rounded to half."""
    if precision != "bf16x3p":
        pytest.skip("one arithmetic mode is enough for the dtype plumbing")
    m, cfg, specs, P = build("micro", dev)
    B, N, F = fill.SHAPES["micro"]
    _, h = fill.inputs("micro", B, N, F, cfg["n_mels"])
    zlat = fill.normal("micro/latent", (B, N), 0.6)
    mh = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
    mh.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    mh = mh.to(dev).half()
    with torch.no_grad():
        xh, _ = mh.reverse(T(zlat, dev).half(), T(h, dev).half())
        m32 = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
        m32.load_state_dict({k: v.float() for k, v in mh.state_dict().items()})
        x32, _ = m32.to(dev).reverse(T(zlat, dev).half().float(), T(h, dev).half().float())
        y = mh.infer(T(h, dev).half()[0], sigma=0.6)
    assert xh.dtype == torch.float16 and y.dtype == torch.float16 and y.shape == (F * cfg["hop_size"],)
    assert torch.equal(xh, x32.half())


def test_waveflow_after_remove_weight_norms(dev, precision):
    """inference.py:19-22 folds the weight norm away before synthesis (model.apply(remove_weight_norms)); the engine then receives
    plain weights (NULL weight_g entries) and must give the same forward / inverse, and still train."""
    name = "wf8"
    cfg = fill.WF_CONFIGS[name]
    B, N, F = fill.WF_SHAPES[name]
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    with torch.no_grad():
        z0, ld0 = m(T(audio, dev), T(mel, dev))
    m.apply(cm.remove_weight_norms)
    assert not any(n.endswith("weight_g") for n, _ in m.named_parameters())
    with torch.no_grad():
        z1, ld1 = m(T(audio, dev), T(mel, dev))
        x1, _ = m.reverse(z1, T(mel, dev))
    assert float((z1 - z0).abs().max()) < 1e-5 and float((ld1 - ld0).abs().max()) < 1e-3
    assert float((x1 - T(audio, dev)).abs().max()) < Z_ATOL
    z2, ld2 = m(T(audio, dev), T(mel, dev))
    cm.WaveGlowLoss(fill.SIGMA)(z2, ld2).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


@pytest.mark.parametrize("B,Tn,depth,aux,ch,planned", [(3, 1000, 8, 80, 256, True), (5, 130, 8, 80, 256, True), (2, 2000, 4, 80, 256, True),
                                                       (4, 2000, 2, 80, 256, True), (1, 300, 8, 80, 256, True), (2, 700, 2, 80, 256, True),
                                                       (1, 64, 8, 80, 256, True), (2, 500, 3, 80, 256, True), (2, 512, 8, 1500, 256, True),
                                                       (3, 700, 5, 3659, 256, True), (2, 500, 3, 80, 64, False), (24, 2000, 8, 80, 256, True),
                                                       (6, 1024, 8, 1500, 256, True)])
def test_weight_gradient_kernel_plans_vs_oracle(dev, precision, B, Tn, depth, aux, ch, planned):
    """wgrad16t_kernel (one workgroup per CU, planned phases, wg_wgrad16t.h) at the shipped WN width over several (batch, length, depth,
    conditioning width) combinations: different K ranges, part counts and phase shapes of the two planners -- the two-phase plan of the
    headline shape; the ROUNDS plan for a layer count that does not divide the 8 XCDs and for rows of tiles wider than an XCD (WSRGlow's
    3 659 conditioning channels: 35 column tiles in sub-sets of 7); the headline shape's own K = 24 x 2000 columns; six items of 1024 columns with 1 500 conditioning channels, whose conditioning gradient (12 row
    tiles x 48 column tiles, plane rows that do not divide by the 8 XCDs) walks the XCD-column tile order (ConvGemm16sArgs::xcd_items < 0) -- against the oracle, and a case WITHOUT a plan (a 64-channel WN: its
    products have 128 rows, the kernel's tiles 256), which must take the two-workgroup kernel and agree as well."""
    if precision != "bf16x3p":
        pytest.skip("the grouped weight-gradient launches exist in the S-plane mode only")
    from constant_memory_waveglow_amd import _lib
    wn = dict(in_channels=4, aux_channels=aux, residual_channels=ch, dilation_channels=ch, skip_channels=ch, depth=depth, radix=3)
    specs = fill.wn_param_specs("F.", 4, aux, ch, ch, ch, depth, 3)
    tag = "coupling/wgt%d_%d_%d" % (B, Tn, depth) + ("" if aux == 80 else "_%d" % aux) + ("" if ch == 256 else "_c%d" % ch)
    P = fill.fill_params(specs, tag + "/")
    x = fill.uniform(tag + "/x", (B, 8, Tn))
    y = fill.normal(tag + "/y", (B, aux, Tn))
    gz = fill.normal(tag + "/gz", x.shape)
    gls = fill.normal(tag + "/gls", (B, 4, Tn))
    z_ref, _ = orc.coupling_apply(wn, fill.table(specs, P), x, y)
    ref = orc.coupling_backward(wn, fill.table(specs, P), z_ref, y, gz, gls)
    blk = cm.AffineCouplingBlock(cm.WN, True, zero_init=False, **wn)
    blk.load_state_dict({n: torch.from_numpy(v) for n, v in P.items()})
    blk = blk.to(dev)
    xt, yt = T(x, dev).requires_grad_(True), T(y, dev).requires_grad_(True)
    before = _lib.lib().wg_stat_wgrad16t_launches()
    z, ls = blk(xt.clone(), yt)
    ((z * T(gz, dev)).sum() + (ls * T(gls, dev)).sum()).backward()
    torch.cuda.synchronize()
    assert (_lib.lib().wg_stat_wgrad16t_launches() - before == 1) == planned
    # (the forward too: six items of 1024 columns with 1 500 conditioning channels is the one shape here whose gate convs are cut along K while
    # the skip path runs in its rank form -- the end conv must then read the gate planes, not partial rows nobody wrote)
    assert np.abs(npy(z) - z_ref).max() < Z_ATOL
    named = dict(blk.named_parameters())
    for i, (n, _, _) in enumerate(specs):
        assert relmax(npy(named[n].grad), ref["grads"][i]) < GRAD_RTOL, n
    assert relmax(npy(xt.grad), ref["dx"]) < GRAD_RTOL and relmax(npy(yt.grad), ref["dy"]) < GRAD_RTOL
