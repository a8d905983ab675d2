"""-m gpu: the training-step caller (SURVEY.md 8 rows a17 / a18 / e) on the device.

* the four scalars the reference logs every step (model/lightning.py:58-64) out of the loss kernels, against torch on the same
  z / logdet;
* FlowTrainer (WaveGlow and WSRGlow) against the autograd path of the same model;
* the data-parallel path on ONE GPU: a 1-rank RCCL group with the collectives forced on, so that the per-flow gradient events
  recorded inside wg_train_step, the communication side stream, the asynchronous all-reduce and the per-bucket Adam step behind it
  all execute on hardware -- gradients, metrics and updated weights must be bit-equal to the non-distributed path
  (what DDP's bucket hooks do for the reference, train.py:51-53,77).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import fill
import constant_memory_waveglow_amd as cm
from constant_memory_waveglow_amd import engine
from constant_memory_waveglow_amd.parallel import FlatAdam, FlowTrainer, METRIC_NAMES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu suite needs the MI355X"
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def build(name, dev, mem_eff=True):
    cfg = fill.CONFIGS[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    kw = dict(cfg)
    m = cm.WaveGlow(memory_efficient=mem_eff, bias=kw.pop("bias", False), **kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    return m.to(dev), cfg


def build_wsr(name, dev):
    cfg = fill.CONFIGS[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    P.update(fill.wsr_tables(name + "/"))
    m = cm.WSRGlow(upsample_rate=fill.WSR_RATE[name], memory_efficient=True, bias=False, **fill.WSR_KW)
    sd = {k: torch.from_numpy(v) for k, v in P.items()}
    sd["window"] = torch.hann_window(16)
    m.load_state_dict(sd)
    return m.to(dev), cfg


def torch_metrics(z, logdet, loss):
    """model/lightning.py:58-64, literally"""
    return [float(logdet.sum() / z.numel()), float(z.mean()), float(z.std()), float(loss)]


@pytest.mark.parametrize("B,N", [(1, 2), (3, 4000), (24, 16000), (5, 1237)])
@pytest.mark.parametrize("mean", [True, False])
def test_logged_scalars_vs_torch(dev, B, N, mean):
    """wg_nll_loss's metrics vector == [logdet.sum()/z.numel(), z.mean(), z.std(), loss] as torch evaluates them."""
    z = T(fill.normal("met/z%d_%d" % (B, N), (B, N), 0.7) + 0.013, dev)
    ld = T(fill.normal("met/ld%d_%d" % (B, N), (B,), 300.0), dev)
    m = torch.full((4,), float("nan"), device=dev)
    loss = engine.nll_loss(z, ld, 0.7, mean, metrics=m)
    z64, ld64 = z.double(), ld.double()
    want_loss = float((0.5 * (z64 * z64).sum(1) / 0.49 - ld64).mean() / (N if mean else 1))
    assert abs(float(loss) - want_loss) <= 1e-6 * max(1.0, abs(want_loss))
    want = [float(ld64.sum() / z.numel()), float(z64.mean()), float(z64.std()), want_loss]
    got = m.tolist()
    for g, w, n in zip(got, want, METRIC_NAMES):
        assert abs(g - w) <= 2e-6 * max(1.0, abs(w)), (n, g, w)
    ref32 = torch_metrics(z, ld, loss)                                   # and what fp32 torch itself logs
    for g, w, n in zip(got[:3], ref32[:3], METRIC_NAMES):
        assert abs(g - w) <= 1e-5 * max(1.0, abs(w)), (n, g, w)
    assert torch.equal(engine.training_metrics(z, ld, 0.7, mean), m)


def test_trainer_reports_the_logged_scalars(dev):
    m, cfg = build("c1", dev)
    B, N, F = fill.SHAPES["c1"]
    audio, h = fill.inputs("c1", B, N, F, cfg["n_mels"])
    tr = FlowTrainer(m, fill.SIGMA)
    loss, z, logdet = tr.step(T(audio, dev), T(h, dev))
    want = torch_metrics(z, logdet, loss)
    got = tr.metrics_dict()
    assert list(got) == ["logdet", "z_mean", "z_std", "loss"]            # the keys of log_dict / log upstream
    for (n, g), w in zip(got.items(), want):
        assert abs(g - w) <= 1e-5 * max(1.0, abs(w)), (n, g, w)
    assert got["loss"] == float(loss)


def _snapshot(tr):
    return tr.fg.flat.clone()


@pytest.fixture(scope="module")
def one_rank_rccl(dev):
    """a 1-rank RCCL ("nccl") process group for the tests below; torn down afterwards"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dist.group.WORLD
    dist.destroy_process_group()


@pytest.mark.parametrize("name,mem_eff", [("micro", True), ("micro_bias", True), ("micro_bias", False), ("c1", True), ("c1", False)])
def test_rccl_path_one_rank_is_bit_equal(dev, one_rank_rccl, name, mem_eff):
    """FlowTrainer.step through the collective path (events from inside wg_train_step, side stream, async all-reduce) == the
    plain step, bit for bit: gradients, z, logdet, loss and the reduced metrics."""
    m0, cfg = build(name, dev, mem_eff)
    m1, _ = build(name, dev, mem_eff)
    B, N, F = fill.SHAPES[name]
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    plain = FlowTrainer(m0, fill.SIGMA, force_collectives=False)
    coll = FlowTrainer(m1, fill.SIGMA, force_collectives=True)
    assert plain.events is None and coll.events is not None and len(coll.events) == cfg["flows"] + 1
    for rep in range(3):
        l0, z0, ld0 = plain.step(x, ht)
        l1, z1, ld1 = coll.step(x, ht)
        torch.cuda.synchronize()
        assert coll.sync._comm is not None                               # the side stream was used
        assert torch.equal(z0, z1) and torch.equal(ld0, ld1) and torch.equal(l0, l1)
        assert torch.equal(plain.fg.flat, coll.fg.flat), rep             # every bucket and the metric tail
        assert torch.isfinite(coll.fg.flat).all()
    for p0, p1 in zip(m0.parameters(), m1.parameters()):
        assert torch.equal(p0.grad, p1.grad)


def test_rccl_path_with_adam_is_bit_equal(dev, one_rank_rccl):
    """With FlatAdam attached the optimizer step of a bucket runs on the communication stream right behind that bucket's
    all-reduce (after_bucket): three steps must leave bit-identical weights and optimizer state on both paths."""
    m0, cfg = build("c1", dev)
    m1, _ = build("c1", dev)
    B, N, F = fill.SHAPES["c1"]
    audio, h = fill.inputs("c1", B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    plain = FlowTrainer(m0, fill.SIGMA, force_collectives=False)
    coll = FlowTrainer(m1, fill.SIGMA, force_collectives=True)
    o0, o1 = FlatAdam(plain, lr=1e-3), FlatAdam(coll, lr=1e-3)
    losses = []
    for _ in range(3):
        l0, _, _ = plain.step(x, ht)
        l1, _, _ = coll.step(x, ht)
        torch.cuda.synchronize()
        assert torch.equal(l0, l1)
        assert torch.equal(o0.flat, o1.flat) and torch.equal(o0.exp_avg, o1.exp_avg) and torch.equal(o0.exp_avg_sq, o1.exp_avg_sq)
        losses.append(float(l1))
    assert losses[2] < losses[0]
    for p0, p1 in zip(m0.parameters(), m1.parameters()):
        assert torch.equal(p0, p1)


@pytest.mark.parametrize("name", ["wsr", "wsr3"])
def test_wsrglow_trainer_matches_autograd(dev, one_rank_rccl, name):
    """WSRGlow on FlowTrainer (per-flow buckets + a front-end bucket for the two embedding tables) == the autograd path of the
    same model, and == itself through the forced collective path."""
    m, cfg = build_wsr(name, dev)
    B, N, F = fill.SHAPES[name]
    audio, c = fill.wsr_inputs(name, B, N, fill.WSR_RATE[name])
    x = T(audio, dev)
    z, logdet = m(x, T(c, dev))
    cm.WaveGlowLoss(1.0)(z, logdet).backward()
    ga = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    tr = FlowTrainer(m, 1.0, force_collectives=False)
    assert len(tr.fg.bucket_ranges) == cfg["flows"] + 2
    ct = T(c, dev)
    loss, z2, ld2 = tr.step(x, ct)
    assert float(ct.abs().max()) <= 1.0                                  # clipped in place, as upstream (wsrglow.py:38)
    assert torch.equal(z, z2) and torch.equal(logdet, ld2)
    tables = ("mu_enc.1.weight", "angle_embed.embed.weight")
    for n, p in m.named_parameters():
        if n in tables:
            # embedding-table gradients are scatter-adds through LDS float atomics: the summation order inside a slice is not fixed
            # (like torch's own CUDA embedding backward), so they repeat to rounding, not bit for bit
            assert float((ga[n] - p.grad).abs().max()) <= 1e-5 * float(ga[n].abs().max()), n
        else:
            assert torch.equal(ga[n], p.grad), n
    m1, _ = build_wsr(name, dev)
    coll = FlowTrainer(m1, 1.0, force_collectives=True)
    assert len(coll.events) == cfg["flows"] + 2
    for _ in range(2):
        l1, z1, _ = coll.step(x, T(c, dev))
        torch.cuda.synchronize()
    assert torch.equal(z1, z)
    lo, hi = tr.fg.bucket_ranges[cfg["flows"] + 1]                       # the embedding-table bucket: equal to rounding (see above)
    assert torch.equal(coll.fg.flat[:lo], tr.fg.flat[:lo]) and torch.equal(coll.fg.tail, tr.fg.tail)
    assert float((coll.fg.flat[lo:hi] - tr.fg.flat[lo:hi]).abs().max()) <= 1e-5 * float(tr.fg.flat[lo:hi].abs().max())


def test_autograd_model_gradients_through_one_rank_rccl(dev, one_rank_rccl):
    """WaveFlow trains through autograd; its data-parallel step is GradSync.all_reduce_params (one flat buffer, one collective).
    On a 1-rank RCCL group the mean all-reduce must hand every gradient back unchanged, bit for bit."""
    from constant_memory_waveglow_amd.parallel import GradSync
    name = "wf8"
    cfg = fill.WF_CONFIGS[name]
    B, N, F = fill.WF_SHAPES[name]
    P = fill.fill_params(fill.waveflow_param_specs(cfg), name + "/")
    audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
    m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    z, logdet = m(T(audio, dev), T(mel, dev))
    cm.WaveGlowLoss(fill.SIGMA)(z, logdet).backward()
    before = {n: p.grad.clone() for n, p in m.named_parameters()}
    sync = GradSync(force_collectives=True)
    sync.all_reduce_params(list(m.parameters()))
    sync.broadcast_params(list(m.parameters()))
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        assert torch.equal(before[n], p.grad), n


def test_trainer_with_bias_matches_autograd(dev):
    """WaveGlow(bias=True): FlowTrainer.step (wg_train_step, the biases in their flow's gradient bucket) leaves in p.grad what
    loss.backward() through the autograd node leaves, for every parameter including the biases."""
    m0, cfg = build("micro_bias", dev)
    m1, _ = build("micro_bias", dev)
    B, N, F = fill.SHAPES["micro_bias"]
    audio, h = fill.inputs("micro_bias", B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    tr = FlowTrainer(m0, fill.SIGMA)
    loss, z, logdet = tr.step(x, ht)
    z1, ld1 = m1(x, ht)
    l1 = cm.WaveGlowLoss(fill.SIGMA)(z1, ld1)
    l1.backward()
    assert torch.equal(z, z1) and abs(float(loss) - float(l1)) < 1e-6
    n_bias = 0
    for (n, p0), (_, p1) in zip(m0.named_parameters(), m1.named_parameters()):
        assert p0.grad is not None and p1.grad is not None, n
        scale = max(float(p1.grad.abs().max()), 1e-30)
        assert float((p0.grad - p1.grad).abs().max()) / scale < 1e-5, n
        n_bias += n.endswith(".bias") and ".F." in n
    assert n_bias == cfg["flows"] * (2 + 2 * cfg["depth"] + 1)


def test_frozen_weight_v_still_gets_weight_g_gradient(dev):
    """weight_v frozen, weight_g trainable (a fine-tuning set-up the reference's autograd handles): the finalisation must still
    produce dg, and must not touch a dv buffer that does not exist."""
    m, cfg = build("micro", dev)
    B, N, F = fill.SHAPES["micro"]
    audio, h = fill.inputs("micro", B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    z, ld = m(x, ht)
    cm.WaveGlowLoss(fill.SIGMA)(z, ld).backward()
    want = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    for n, p in m.named_parameters():
        if n.endswith("weight_v"):
            p.requires_grad_(False)
    z, ld = m(x, ht)
    cm.WaveGlowLoss(fill.SIGMA)(z, ld).backward()
    for n, p in m.named_parameters():
        if n.endswith("weight_v"):
            assert p.grad is None, n
        else:
            assert p.grad is not None and torch.equal(p.grad, want[n]), n


def test_engine_follows_the_tensors_device_not_the_current_one(dev):
    """The C ABI is handed the stream of the tensors' device and runs with that device current (engine.on_device).  With one GPU
    this can only check the plumbing: a step issued from inside another stream context lands on that stream."""
    m, cfg = build("micro", dev)
    B, N, F = fill.SHAPES["micro"]
    audio, h = fill.inputs("micro", B, N, F, cfg["n_mels"])
    x, ht = T(audio, dev), T(h, dev)
    with torch.no_grad():
        z0, _ = m(x, ht)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            z1, _ = m(x, ht)
        side.synchronize()
    assert torch.equal(z0, z1)
    with pytest.raises(cm.WgError):
        engine.require_device(x, ht.cpu())
