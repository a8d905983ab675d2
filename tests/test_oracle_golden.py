"""Pins the CPU oracle (oracle/wg_oracle.c) to the golden vectors the upstream reference produced
(tests/golden/make_golden.py).  CPU only.  Tolerances: the reference's own fp32 result sits ~1e-6 (z),
~3e-6 of each tensor's max (grads) and ~1e-7 relative (logdet) from fp64."""
import os

import numpy as np
import pytest

import fill
from oracle import wg_oracle as orc
from make_golden import COUPLING_CASES

Z_ATOL = 5e-6
GRAD_RTOL = 2e-5      # relative to each tensor's max-abs
LOGDET_RTOL = 2e-6   # plus 5e-8 per audio sample: logdet is a sum of ~N terms that partly cancel


def _logdet_close(a, b, N):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= LOGDET_RTOL * np.abs(b) + 5e-8 * N))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _relmax(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-30))


@pytest.mark.parametrize("name", ["micro", "micro_bias", "micro_r5", "c1", "c2", "wsr_like"])
@pytest.mark.parametrize("double", [False, True])
def test_model_step_matches_reference(golden_dir, name, double):
    if name == "c2" and double:
        pytest.skip("fp64 C2 step takes ~20 s; the fp32 run already pins it")
    g = _load(golden_dir, "model_%s.npz" % name)
    cfg = fill.CONFIGS[name]
    B, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    tab = fill.table(specs, P)
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    oc = orc.make_config(**cfg)
    assert orc.param_count(oc) == len(specs)
    r = orc.train_step(oc, tab, audio, h, fill.SIGMA, need_dh=True, double=double)
    assert np.abs(r["z"] - g["z"]).max() < Z_ATOL
    assert _logdet_close(r["logdet"], g["logdet"], N)
    assert abs(r["loss"] - float(g["loss"])) < 1e-6
    assert _relmax(r["dh"], g["dh"]) < GRAD_RTOL
    for i, (n, _, _) in enumerate(specs):
        gr = r["grads"][i].ravel()
        scale = max(float(g["grad_max"][i]), 1e-30)
        nh = min(gr.size, g["grad_head"].shape[1])
        assert np.abs(gr[:nh] - g["grad_head"][i][:nh]).max() / scale < GRAD_RTOL, n
        nrm = np.sqrt((gr.astype(np.float64) ** 2).sum())
        assert abs(nrm - g["grad_norm"][i]) <= 1e-4 * g["grad_norm"][i] + 1e-12, n
        if "grad::" + n in g:
            assert _relmax(r["grads"][i], g["grad::" + n]) < GRAD_RTOL, n


@pytest.mark.parametrize("name", ["micro", "micro_bias", "micro_r5", "c1"])
def test_model_inverse_matches_reference(golden_dir, name):
    g = _load(golden_dir, "model_%s.npz" % name)
    cfg = fill.CONFIGS[name]
    B, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    tab = fill.table(specs, fill.fill_params(specs, name + "/"))
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    oc = orc.make_config(**cfg)
    x, ld = orc.inverse(oc, tab, g["z"], h)
    assert np.abs(x - g["x_inv"]).max() < Z_ATOL
    assert np.abs(x - audio).max() < 1e-5                       # reverse(forward(x)) == x
    assert _logdet_close(ld, g["logdet_inv"], N)
    assert _logdet_close(ld, -g["logdet"], N)
    zlat = fill.normal(name + "/latent", (B, N), fill.SIGMA)
    xs, _ = orc.inverse(oc, tab, zlat, h)
    assert np.abs(xs - g["x_from_latent"]).max() < 2e-5 * max(1.0, float(np.abs(g["x_from_latent"]).max()))


def test_loss_value():
    z = fill.normal("loss/z", (3, 64))
    ld = fill.normal("loss/ld", (3,))
    want = float(np.mean(0.5 * (z.astype(np.float64) ** 2).sum(1) / 0.49 - ld) / 64)
    assert abs(orc.loss(z, ld, 0.7) - want) < 1e-6


@pytest.mark.parametrize("c", [2, 4, 8])
@pytest.mark.parametrize("shape", [(1, 64), (3, 200)])
@pytest.mark.parametrize("rev", [False, True])
def test_invconv_block(golden_dir, c, shape, rev):
    g = _load(golden_dir, "block_invconv.npz")
    B, T = shape
    tag = "invconv/c%d_b%d_t%d" % (c, B, T)
    k = tag + ("/rev" if rev else "/fwd")
    W = fill.orthogonal(tag + "/W", c)
    x = fill.uniform(tag + "/x", (B, c, T))
    gz = fill.normal(tag + "/gz", (B, c, T))
    y, ld = (orc.invconv_reverse if rev else orc.invconv_forward)(W, x)
    assert np.abs(y - g[k + "/y"]).max() < 2e-6
    assert abs(float(ld) - float(g[k + "/logdet"])) < 1e-5 * max(1.0, abs(float(g[k + "/logdet"])))
    xr, dx, dW = orc.invconv_backward(W, y, gz, 0.37, reverse=rev)
    assert np.abs(xr - x).max() < 2e-6                          # input re-materialised from the output
    assert _relmax(dx, g[k + "/dx"]) < GRAD_RTOL
    assert _relmax(dW, g[k + "/dW"]) < GRAD_RTOL


def test_invconv_negative_det_gives_nan():
    W = fill.orthogonal("negdet", 4)
    W[:, 0] = -W[:, 0]
    _, ld = orc.invconv_forward(W, fill.uniform("negdet/x", (1, 4, 8)))
    assert np.isnan(ld)                                          # torch.logdet semantics (efficient_modules.py:38)


@pytest.mark.parametrize("cname", list(COUPLING_CASES))
@pytest.mark.parametrize("rev", [False, True])
def test_coupling_block(golden_dir, cname, rev):
    g = _load(golden_dir, "block_coupling.npz")
    cs = COUPLING_CASES[cname]
    tag = "coupling/" + cname
    k = tag + ("/rev" if rev else "/fwd")
    wn = dict(in_channels=cs["c"] // 2, aux_channels=cs["aux"], residual_channels=cs["wn"], dilation_channels=cs["wn"],
              skip_channels=cs["wn"], depth=cs["depth"], radix=3)
    specs = fill.wn_param_specs("F.", cs["c"] // 2, cs["aux"], cs["wn"], cs["wn"], cs["wn"], cs["depth"], 3)
    tab = fill.table(specs, fill.fill_params(specs, tag + "/"))
    x = fill.uniform(tag + "/x", (cs["B"], cs["c"], cs["T"]))
    y = fill.normal(tag + "/y", (cs["B"], cs["aux"], cs["T"]))
    gz = fill.normal(tag + "/gz", x.shape)
    gls = fill.normal(tag + "/gls", (cs["B"], cs["c"] // 2, cs["T"]))
    z, ls = orc.coupling_apply(wn, tab, x, y, reverse=rev)
    assert np.abs(z - g[k + "/z"]).max() < 1e-5
    assert np.abs(ls - g[k + "/log_s"]).max() < 1e-5
    r = orc.coupling_backward(wn, tab, z, y, gz, gls, reverse=rev)
    assert np.abs(r["x"] - x).max() < 1e-5
    assert _relmax(r["dx"], g[k + "/dx"]) < GRAD_RTOL
    assert _relmax(r["dy"], g[k + "/dy"]) < GRAD_RTOL
    for i, (n, _, _) in enumerate(specs):
        gr = r["grads"][i]
        nrm = np.sqrt((gr.astype(np.float64) ** 2).sum())
        assert abs(nrm - g[k + "/grad_norm"][i]) <= 2e-4 * g[k + "/grad_norm"][i] + 1e-12, n
        if k + "/grad::" + n in g:
            assert _relmax(gr, g[k + "/grad::" + n]) < GRAD_RTOL, n


def test_reverse_mode_architecture(golden_dir):
    """WaveGlow(reverse_mode=True) (model/base.py:20-28 double swap): forward, backward and inverse."""
    name = "micro"
    g = _load(golden_dir, "model_micro_rm.npz")
    cfg = fill.CONFIGS[name]
    B, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    tab = fill.table(specs, fill.fill_params(specs, name + "/"))
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    oc = orc.make_config(**cfg)
    r = orc.train_step(oc, tab, audio, h, fill.SIGMA, need_dh=True, reverse_mode=True)
    assert np.abs(r["z"] - g["z"]).max() < Z_ATOL
    assert _logdet_close(r["logdet"], g["logdet"], N)
    assert abs(r["loss"] - float(g["loss"])) < 1e-6
    assert _relmax(r["dh"], g["dh"]) < GRAD_RTOL
    for i, (n, _, _) in enumerate(specs):
        assert _relmax(r["grads"][i], g["grad::" + n]) < GRAD_RTOL, n
    x, ld = orc.inverse(oc, tab, g["z"], h, reverse_mode=True)
    assert np.abs(x - g["x_inv"]).max() < Z_ATOL and np.abs(x - audio).max() < 1e-5
    assert _logdet_close(ld, g["logdet_inv"], N)
    # it is a different function from the reverse_mode=False model
    assert np.abs(g["z"] - _load(golden_dir, "model_micro.npz")["z"]).max() > 1e-2


# ---- WSRGlow ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["wsr", "wsr3"])
@pytest.mark.parametrize("double", [False, True])
def test_wsrglow_matches_reference(golden_dir, name, double):
    """The oracle's conditioning front-end + flow stack against the reference's WSRGlow (model/wsrglow.py) run by
    make_golden.wsrglow_fixture: quantiser decisions, |STFT|, z, logdet, loss, all gradient norms, both embedding-table grads.
    "wsr" = upsample_rate 2 (configs/wsrglow_vctk_2x.json), "wsr3" = rate 3 (wsrglow_vctk_3x.json: 24 squeezed channels)."""
    cfg = fill.CONFIGS[name]
    B, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    tabs = fill.wsr_tables(name + "/")
    audio, c = fill.wsr_inputs(name, B, N, fill.WSR_RATE[name])
    G = np.load(os.path.join(golden_dir, "model_%s.npz" % name))
    cond, mi, ai = orc.wsr_cond(c, tabs["mu_enc.1.weight"], tabs["angle_embed.embed.weight"], double=double, return_idx=True)
    assert cond.shape == (B, orc.WSR_COND, F)
    assert np.array_equal(mi, G["mu_idx"]) and np.array_equal(ai, G["ang_idx"])
    assert np.abs(cond[:, 3200:3209] - G["mag"]).max() < 2e-6
    assert np.abs(cond[:, ::97, :4] - G["cond_head"]).max() < 2e-6
    assert abs(np.sqrt((cond.astype(np.float64) ** 2).sum()) - float(G["cond_norm"])) < 1e-4 * float(G["cond_norm"])
    r = orc.train_step(orc.make_config(**cfg), fill.table(specs, P), audio, cond, 1.0, need_dh=True, double=double)
    assert np.abs(r["z"] - G["z"]).max() < 1e-5
    # (the fp32 oracle sums 3659-term conditioning products sequentially: its logdet sits up to 1e-7 per sample from the fp64 run,
    # which agrees with the reference to 7e-6 here)
    assert _logdet_close(r["logdet"], G["logdet"], N if double else 2 * N)
    assert abs(r["loss"] - float(G["loss"])) < 1e-6
    gn = np.array([np.sqrt((g.astype(np.float64) ** 2).sum()) for g in r["grads"]])
    assert np.all(np.abs(gn - G["grad_norm"]) <= 2e-5 * G["grad_norm"] + 1e-12)
    dmu, dang = orc.wsr_cond_backward(c, r["dh"], double=double)
    for n, g in (("mu_enc.1.weight", dmu), ("angle_embed.embed.weight", dang)):
        assert np.abs(g - G["grad::" + n]).max() < 2e-5 * np.abs(G["grad::" + n]).max(), n
    x, _ = orc.inverse(orc.make_config(**cfg), fill.table(specs, P), G["z"], cond, double=double)
    assert np.abs(x - G["x_inv"]).max() < 1e-5


def test_wsr_cond_edge_cases():
    """Clip, silence and the shortest input: c beyond [-1,1] behaves as the clipped signal; an all-zero frame has |STFT| = 0 and
    the phase index of angle 0; L = 8 is one frame whose reflect padding reads c[4..1] and c[6..3]."""
    mu_w = fill.normal("edge/mu", (256, 400))
    ang_w = fill.normal("edge/ang", (120, 50))
    c = fill.uniform("edge/c", (2, 8), -2.0, 2.0)
    a, mi, ai = orc.wsr_cond(c, mu_w, ang_w, return_idx=True)
    b = orc.wsr_cond(np.clip(c, -1, 1), mu_w, ang_w)
    assert a.shape == (2, orc.WSR_COND, 1) and np.array_equal(a, b)
    assert mi.min() >= 0 and mi.max() <= 255 and ai.min() >= 0 and ai.max() <= 119
    z, mi, ai = orc.wsr_cond(np.zeros((1, 64), np.float32), mu_w, ang_w, return_idx=True)
    assert np.all(mi == 128) and np.all(ai == 59) and np.all(z[:, 3200:3209] == 0)
    assert np.array_equal(z[0, :400, 0], mu_w[128]) and np.array_equal(z[0, 3209:3259, 0], ang_w[59])
    with pytest.raises(RuntimeError):
        orc.wsr_cond(np.zeros((1, 12), np.float32), mu_w, ang_w)       # L must be a multiple of 8


# ---- WaveFlow ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["wf8", "wf8c"])
def test_waveflow_reverse_mode_matches_reference(golden_dir, name):
    """WaveFlow(reverse_mode=True) (model/base.py:20-28; make_golden.waveflow_rm_fixture): `forward` = the row loop, `reverse` = the parallel
    map, the 1x1 convs swapped as well -- for the oracle that is the SAME two functions with the matrices W^-1 in the parameter table."""
    from oracle import wf_oracle as wfo
    cfg = fill.WF_CONFIGS[name]
    B, N, F = fill.WF_SHAPES[name]
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    G = np.load(os.path.join(golden_dir, "model_%s_rm.npz" % name))
    tab = fill.table(specs, P)
    mix = [i for i, (n, _, _) in enumerate(specs) if "invconv1x1" in n]
    for i in mix:
        tab[i] = np.linalg.inv(tab[i][:, :, 0].astype(np.float64)).astype(np.float32)[:, :, None]
    oc = wfo.make_config(**cfg)
    z, ld = wfo.inverse(oc, tab, audio, mel)
    assert np.abs(z - G["z"]).max() < 2e-5                        # (the row loop amplifies rounding: 2e-5 as for the model's own inverse)
    assert _logdet_close(ld, G["logdet"], N)
    r = wfo.train_step(oc, tab, G["z"], mel, fill.SIGMA, need_dmel=True)
    assert np.abs(r["z"] - G["x_rev"]).max() < 2e-5 and np.abs(r["z"] - audio).max() < 5e-5
    assert _logdet_close(r["logdet"], G["logdet_rev"], N)
    assert abs(r["loss"] - float(G["loss"])) < 1e-6
    assert np.abs(r["dmel"] - G["dmel"]).max() < 1e-5 * np.abs(G["dmel"]).max()
    for i, (n, _, _) in enumerate(specs):
        g = r["grads"][i].astype(np.float64)
        if n.endswith("start.weight_v"):
            continue
        if i in mix:                                              # dL/dW = -M^T (dL/dM) M^T with M = W^-1
            mt = tab[i][:, :, 0].astype(np.float64).T
            g = -(mt @ g[:, :, 0] @ mt)[:, :, None]
            assert np.abs(g - G["grad::" + n]).max() <= 2e-5 * np.abs(G["grad::" + n]).max(), n
        gn = float(np.sqrt((g ** 2).sum()))
        assert abs(gn - float(G["grad_norm"][i])) <= 2e-5 * float(G["grad_norm"][i]) + 1e-12, n


@pytest.mark.parametrize("name", ["wf8", "wf64", "wf8c", "wf64c", "wf8b", "wf64b"])
@pytest.mark.parametrize("double", [False, True])
def test_waveflow_matches_reference(golden_dir, name, double):
    """oracle/wf_oracle.c against the reference's WaveFlow (model/waveflow.py) run by make_golden.waveflow_fixture:
    z, logdet, loss, every parameter-gradient norm and head, d loss / d mel, and the row-by-row inverse.
    "wf8c" / "wf64c": use_conv1x1=True (an invertible 1x1 conv over the height axis instead of the flip, waveflow.py:203-206);
    "wf8b" / "wf64b": WN2D(bias=True) (waveflow.py:77), every bias gradient stored in full."""
    from oracle import wf_oracle as wfo
    cfg = fill.WF_CONFIGS[name]
    B, N, F = fill.WF_SHAPES[name]
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    G = np.load(os.path.join(golden_dir, "model_%s.npz" % name))
    oc = wfo.make_config(**cfg)
    assert wfo.param_count(oc) == len(specs)
    r = wfo.train_step(oc, fill.table(specs, P), audio, mel, fill.SIGMA, need_dmel=True, double=double)
    assert np.abs(r["z"] - G["z"]).max() < 2e-6
    assert _logdet_close(r["logdet"], G["logdet"], N)
    assert abs(r["loss"] - float(G["loss"])) < 1e-6
    assert np.abs(r["dmel"] - G["dmel"]).max() < 1e-5 * np.abs(G["dmel"]).max()
    for i, (n, shape, _) in enumerate(specs):
        g = r["grads"][i]
        if n.endswith("start.weight_v"):
            # Conv2d(1, C, 1) under weight norm: w = g * sign(v), the exact gradient w.r.t. v is zero; both sides hold rounding noise
            assert np.abs(g).max() < 1e-6 * np.abs(r["grads"][i - 1]).max(), n
            continue
        gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        assert abs(gn - float(G["grad_norm"][i])) <= 2e-5 * float(G["grad_norm"][i]) + 1e-12, n
        nh = min(g.size, G["grad_head"].shape[1])
        assert np.abs(g.ravel()[:nh] - G["grad_head"][i][:nh]).max() <= 2e-5 * float(G["grad_max"][i]) + 1e-12, n
        if "grad::" + n in G:
            assert np.abs(g - G["grad::" + n]).max() <= 2e-5 * np.abs(G["grad::" + n]).max(), n
    zf, ldf = wfo.forward(oc, fill.table(specs, P), audio, mel, double=double)
    assert np.array_equal(zf, r["z"]) or np.abs(zf - r["z"]).max() < 1e-6
    x, ld = wfo.inverse(oc, fill.table(specs, P), G["z"], mel, double=double)
    itol = 1e-5 if cfg.get("use_conv1x1") else 2e-6          # a 64 x 64 fp32 inverse per flow: the reference's own round trip is ~4e-6 off
    assert np.abs(x - G["x_inv"]).max() < itol and np.abs(x - audio).max() < itol
    assert _logdet_close(ld, G["logdet_inv"], N)


@pytest.mark.parametrize("name", ["wf8", "wf64", "wf8b"])
@pytest.mark.parametrize("double", [False, True])
def test_waveflow_torch_cpu_matches_reference(golden_dir, name, double):
    """oracle/torch_cpu.waveflow_train_step -- the float64-capable checker of the WaveFlow workload at its full size
    (tests/test_gpu_parity.py::test_waveflow_timed_workload_vs_float64_oracle) -- pinned to the reference's own step
    (make_golden.waveflow_fixture): z, logdet, loss, d loss / d mel, every gradient's norm / head, and the batch cut into two worker
    processes giving the same numbers (train_step_parallel with cfg['model'] = 'waveflow')."""
    from oracle import torch_cpu as tc
    cfg = dict(fill.WF_CONFIGS[name], model="waveflow")
    B, N, F = fill.WF_SHAPES[name]
    specs = fill.waveflow_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    G = np.load(os.path.join(golden_dir, "model_%s.npz" % name))
    tab = fill.table(specs, P)
    r = tc.waveflow_train_step(cfg, tab, audio, mel, fill.SIGMA, need_dh=True, double=double)

    def check(r):
        assert np.abs(r["z"] - G["z"]).max() < 2e-6
        assert _logdet_close(r["logdet"], G["logdet"], N)
        assert abs(r["loss"] - float(G["loss"])) < 1e-6
        assert np.abs(r["dh"] - G["dmel"]).max() < 1e-5 * np.abs(G["dmel"]).max()
        for i, (n, shape, _) in enumerate(specs):
            g = r["grads"][i]
            if n.endswith("start.weight_v"):
                assert np.abs(g).max() < 1e-6 * np.abs(r["grads"][i - 1]).max(), n
                continue
            gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
            assert abs(gn - float(G["grad_norm"][i])) <= 2e-5 * float(G["grad_norm"][i]) + 1e-12, n
            nh = min(g.size, G["grad_head"].shape[1])
            assert np.abs(g.ravel()[:nh] - G["grad_head"][i][:nh]).max() <= 2e-5 * float(G["grad_max"][i]) + 1e-12, n
            if "grad::" + n in G:
                assert np.abs(g - G["grad::" + n]).max() <= 2e-5 * np.abs(G["grad::" + n]).max(), n
    check(r)
    if name == "wf8" and double:
        check(tc.train_step_parallel(cfg, tab, audio, mel, fill.SIGMA, workers=2, threads=1, need_dh=True, double=True))


# ---- log-mel conditioner ----------------------------------------------------------------------------

@pytest.mark.parametrize("tag", list(fill.MEL_CASES))
def test_mel_oracle_matches_reference_melspec(golden_dir, tag):
    """oracle/mel_oracle.melspec against the reference's MelSpec class (model/condition.py:7-19) run by make_golden.melspec_fixture:
    nn.ReflectionPad1d and the log are the reference's code, the power spectrogram is torch.stft's, the HTK filterbank is the
    published torchaudio formula (ref_shim.load_melspec).  The power spectrogram pins pad + window + STFT on its own."""
    from oracle import mel_oracle as mo
    G = np.load(os.path.join(golden_dir, "cond_melspec.npz"))
    kw = fill.MEL_KW
    x = fill.mel_input(tag)
    pw = mo.power_spectrogram(x, kw["n_fft"], kw["hop_length"])
    want = G[tag + "/power"].astype(np.float64)
    assert pw.shape == want.shape
    assert np.abs(pw - want).max() <= 2e-5 * want.max()                  # torch's fp32 FFT against the float64 one
    lm = mo.melspec(x, kw["sr"], kw["n_fft"], kw["hop_length"], 0.0, kw["f_max"], kw["n_mels"])
    assert lm.shape == G[tag + "/logmel"].shape
    assert np.abs(lm - G[tag + "/logmel"]).max() < 2e-4                  # log domain; fp32 power sums near the 1e-7 floor


def test_upsampler_oracle_matches_reference(golden_dir):
    """wgo_upsample against WaveGlow._upsample_h of the reference (model/waveglow.py:126-130,210-212)."""
    G = np.load(os.path.join(golden_dir, "block_upsampler.npz"))
    for name in ("micro", "c1"):
        cfg = fill.CONFIGS[name]
        B, N, F = fill.SHAPES[name]
        P = fill.fill_params(fill.model_param_specs(cfg), name + "/")
        _, h = fill.inputs(name, B, N, F, cfg["n_mels"])
        y = orc.upsample(orc.make_config(**cfg), P["upsampler.bias"], P["upsampler.weight_g"], P["upsampler.weight_v"], h, G[name].shape[2])
        assert y.shape == G[name].shape and np.abs(y - G[name]).max() < 2e-6


def test_training_metrics_formula():
    """The four scalars LightModel.training_step logs (model/lightning.py:58-64), evaluated by torch exactly as written there --
    the definition the HIP kernel (wg_nll_loss metrics) is tested against on the GPU."""
    import torch
    z = torch.from_numpy(fill.normal("metrics/z", (3, 1000), 0.7))
    ld = torch.from_numpy(fill.normal("metrics/ld", (3,), 50.0))
    want = [float(ld.sum() / z.numel()), float(z.mean()), float(z.std())]
    z64 = z.double().numpy()
    assert abs(want[2] - np.sqrt(((z64 - z64.mean()) ** 2).sum() / (z64.size - 1))) < 1e-6      # unbiased, over all elements


# ---- the oracle's own sanity ----------------------------------------------------------------------------------------------

def test_mel_filterbank_matches_third_party_numbers(golden_dir):
    """The HTK mel filterbank of oracle/mel_oracle.py against the matrix transformers.audio_utils.mel_filter_bank produced for the
    same arguments (stored by make_golden.melspec_fixture, which also holds the torchaudio stand-in to it): the one part of MelSpec
    that no code under /root/reference defines is pinned to an independent published implementation."""
    from oracle import mel_oracle as mo
    G = np.load(os.path.join(golden_dir, "cond_melspec.npz"))
    kw = fill.MEL_KW
    fb = np.asarray(mo.mel_filterbank(kw["sr"], kw["n_fft"], kw["n_mels"], 0.0, kw["f_max"]), dtype=np.float64)
    want = G["filterbank/transformers"]
    if fb.shape != want.shape:
        fb = fb.T
    assert fb.shape == want.shape == (kw["n_fft"] // 2 + 1, kw["n_mels"])
    assert np.abs(fb - want).max() < 1e-6 and want.max() > 0.9


def test_mel_oracle_properties():
    from oracle import mel_oracle as mo
    sr, n_fft, hop, n_mels = 22050, 1024, 256, 80
    fb = mo.mel_filterbank(sr, n_fft, n_mels, 0.0, 8000.0)
    assert fb.shape == (513, 80) and fb.min() >= 0.0 and fb.max() <= 1.0
    assert np.all(fb[0] == 0.0) and np.all(fb[int(8000 / (11025 / 512)) + 2:] == 0.0)        # nothing at DC, nothing above f_max
    peaks = fb.argmax(0)
    assert np.all(np.diff(peaks) > 0)                                                       # centres increase with the mel index
    t = np.arange(16000) / sr
    x = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)[None]
    m = mo.melspec(x, sr, n_fft, hop, 0.0, 8000.0, n_mels)
    assert m.shape == (1, 80, 63)                                                           # 63 frames for 16000 samples
    centre_hz = 700.0 * (10.0 ** (np.linspace(0, 2595 * np.log10(1 + 8000 / 700), 82)[1:-1] / 2595.0) - 1.0)
    assert abs(centre_hz[m[0, :, 30].argmax()] - 1000.0) < 60.0                             # a 1 kHz tone lights the 1 kHz filter
    assert np.allclose(mo.melspec(np.zeros((1, 4096), np.float32), sr, n_fft, hop, 0.0, 8000.0, n_mels), np.log(1e-7))


def test_stft_decimate_oracle_matches_reference(golden_dir):
    """oracle/mel_oracle.stft_decimate against the reference's STFTDecimate (model/condition.py:60-66) run under the legacy-stft
    wrapper of tests/golden/ref_shim: the WSRGlow configs' conditioner, ratios 2 and 3, a ragged length."""
    from oracle import mel_oracle as mo
    G = np.load(os.path.join(golden_dir, "cond_stftdecimate.npz"))
    for tag, (B, T, r) in fill.DECIMATE_CASES.items():
        x = fill.uniform("decimate/" + tag, (B, T), -0.9, 0.9)
        y = mo.stft_decimate(x, r)
        assert y.shape == G[tag].shape == (B, (T + r - 1) // r)
        assert np.abs(y - G[tag]).max() < 2e-6


@pytest.mark.parametrize("name", ["micro", "micro_bias", "micro_r5", "c1", "c2", "wsr_like"])
def test_torch_cpu_step_matches_reference(golden_dir, name):
    """oracle/torch_cpu.py (the restatement on ATen's CPU kernels that bench.py times as the CPU baseline) against the reference's golden
    fixtures: z <= 1e-6, loss <= 1e-6, every gradient <= 1e-5 of its tensor's max."""
    from oracle import torch_cpu
    g = _load(golden_dir, "model_%s.npz" % name)
    cfg = fill.CONFIGS[name]
    B, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    r = torch_cpu.train_step(cfg, fill.table(specs, P), audio, h, fill.SIGMA, need_dh=True)
    assert np.abs(r["z"] - g["z"]).max() < 1e-6 * max(1.0, float(np.abs(g["z"]).max()))
    assert _logdet_close(r["logdet"], g["logdet"], N)
    assert abs(r["loss"] - float(g["loss"])) < 1e-6
    assert _relmax(r["dh"], g["dh"]) < 1e-5
    for i, (n, _, _) in enumerate(specs):
        gr = r["grads"][i].ravel()
        scale = max(float(g["grad_max"][i]), 1e-30)
        nh = min(gr.size, g["grad_head"].shape[1])
        assert np.abs(gr[:nh] - g["grad_head"][i][:nh]).max() / scale < 1e-5, n
        nrm = np.sqrt((gr.astype(np.float64) ** 2).sum())
        assert abs(nrm - g["grad_norm"][i]) <= 1e-5 * g["grad_norm"][i] + 1e-12, n
        if "grad::" + n in g:
            assert _relmax(r["grads"][i], g["grad::" + n]) < 1e-5, n


@pytest.mark.parametrize("cname", ["d4", "last", "r5d3"])
def test_noncausal_layer_restatement_vs_reference_golden(golden_dir, cname):
    """oracle/torch_cpu.noncausal_layer (model/waveglow.py:41-46) against the reference's own NonCausalLayer output (block_layer.npz)."""
    from make_golden import LAYER_CASES, layer_inputs
    from oracle import torch_cpu
    C, Cd, Cs, radix, dil, last, wn, B, Tn = LAYER_CASES[cname]
    P, x, y = layer_inputs(cname)
    gold = np.load(os.path.join(golden_dir, "block_layer.npz"))
    res, skip = torch_cpu.noncausal_layer(P.get("W.weight_g"), P["W.weight_v"], P.get("W_o.weight_g"), P["W_o.weight_v"], x, y, dil, last)
    assert np.abs(skip - gold[cname + "/skip"]).max() < 2e-6
    if last:
        assert res is None
    else:
        assert np.abs(res - gold[cname + "/res"]).max() < 2e-6


@pytest.mark.parametrize("cname", ["wf8", "wf64", "wf64_short", "wf8b"])
def test_wn2d_restatement_vs_reference_golden(golden_dir, cname):
    """oracle/torch_cpu.wn2d_forward (model/waveflow.py:128-135) against the reference's own WN2D output (block_wn2d.npz)."""
    from make_golden import wn2d_inputs
    from oracle import torch_cpu
    cfg, P, x, y = wn2d_inputs(cname)
    gold = np.load(os.path.join(golden_dir, "block_wn2d.npz"))
    order = ["V.weight_g", "V.weight_v", "start.weight_g", "start.weight_v"]
    for i in range(8):
        order += ["layers.%d.W.weight_g" % i, "layers.%d.W.weight_v" % i, "layers.%d.W_o.weight_g" % i, "layers.%d.W_o.weight_v" % i]
    order.append("end.weight")
    if cfg.get("bias"):
        order += ["V.bias", "start.bias"] + [n for i in range(8) for n in ("layers.%d.W.bias" % i, "layers.%d.W_o.bias" % i)] + ["end.bias"]
    ls, t = torch_cpu.wn2d_forward([P[k] for k in order], cfg["n_group"], x, y)
    assert np.abs(ls - gold[cname + "/log_s"]).max() < 5e-6 and np.abs(t - gold[cname + "/t"]).max() < 5e-6
    # ... and its gradients (autograd over the restatement) against what the reference's autograd gave for the same scalar
    from make_golden import wn2d_seeds
    import torch
    gls, gt = wn2d_seeds(cname, x.shape)
    pt = [torch.from_numpy(P[k]).requires_grad_(True) for k in order]
    xt, yt = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(y).requires_grad_(True)
    lst, tt = torch_cpu.wn2d_forward_t(pt, cfg["n_group"], xt, yt)
    ((lst * torch.from_numpy(gls)).sum() + (tt * tt * torch.from_numpy(gt)).sum()).backward()
    assert np.abs(xt.grad.numpy() - gold[cname + "/dx"]).max() <= 1e-5 * np.abs(gold[cname + "/dx"]).max()
    assert np.abs(yt.grad.numpy() - gold[cname + "/dy"]).max() <= 1e-5 * np.abs(gold[cname + "/dy"]).max()
    names = list(P)                                            # the fixture's gradient rows follow the state-dict order of the inputs
    for k, q in zip(order, pt):
        i = names.index(k)
        g = q.grad.numpy().ravel()
        scale = max(float(gold[cname + "/grad_max"][i]), 1e-30)
        if k == "start.weight_v":
            assert np.abs(g).max() < 1e-5 * float(gold[cname + "/grad_max"][names.index("start.weight_g")])   # exact zero, rounding noise
            continue
        nh = min(g.size, 8)
        assert np.abs(g[:nh] - gold[cname + "/grad_head"][i][:nh]).max() / scale < 1e-5, k
        assert abs(float(np.sqrt((g.astype(np.float64) ** 2).sum())) - float(gold[cname + "/grad_norm"][i])) <= 1e-5 * float(gold[cname + "/grad_norm"][i]), k

