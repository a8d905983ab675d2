"""-m gpu: the data-parallel trainer with TWO ranks on the ONE GPU of the test box.

RCCL refuses two ranks on one device, so the group is gloo (it moves HIP tensors through host memory): slower, but everything ABOVE the
collective is the product path on hardware with world size 2 -- the flat parameter broadcast from rank 0, per-flow gradient events
recorded inside wg_train_step, the communication side stream, one asynchronous mean all-reduce per bucket in backward order, the metric
tail riding in the last bucket, per-bucket Adam behind each reduction.  What the reference gets from Lightning's DDP
(train.py:51-53,73-78; model/lightning.py:63-64): per-GPU batch = global // world, gradients of the global batch on every rank,
replicas that stay identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fill

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import constant_memory_waveglow_amd as cm
    from constant_memory_waveglow_amd.parallel import FlatAdam, FlowTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    name = "micro"
    cfg = fill.CONFIGS[name]
    _, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")

    def model():
        m = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
        return m.to(dev)

    m = model()
    if rank == 1:                                               # replicas must START identical: rank 1's weights are overwritten by rank 0's
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.01)
    tr = FlowTrainer(m, fill.SIGMA)                             # broadcast_params inside
    assert tr.sync.world == 2 and tr.events is not None
    ref = model()
    for p, r in zip(m.parameters(), ref.parameters()):
        assert torch.equal(p, r), "flat broadcast did not restore rank 0's parameters"
    audio, h = fill.inputs("dp2", 4, N, F, cfg["n_mels"])       # global batch 4, 2 per rank
    mine = slice(2 * rank, 2 * rank + 2)
    x, ht = torch.from_numpy(audio[mine]).to(dev), torch.from_numpy(h[mine]).to(dev)
    loss, z, logdet = tr.step(x, ht)
    torch.cuda.synchronize()
    # the same global batch in ONE process through autograd (no collective in that path)
    xf, hf = torch.from_numpy(audio).to(dev), torch.from_numpy(h).to(dev)
    zf, ldf = ref(xf, hf)
    lf = cm.WaveGlowLoss(fill.SIGMA)(zf, ldf)
    lf.backward()
    worst = 0.0
    for p, r in zip(m.parameters(), ref.parameters()):
        worst = max(worst, float((p.grad - r.grad).abs().max() / r.grad.abs().max().clamp_min(1e-30)))
    met = tr.metrics_dict()
    full = [float(ldf.sum() / zf.numel()), float(zf.mean()), float(lf)]
    emet = max(abs(met["logdet"] - full[0]), abs(met["z_mean"] - full[1]), abs(met["loss"] - full[2]))
    # three optimizer steps: the replicas must stay bit-identical
    FlatAdam(tr, lr=1e-3)
    for _ in range(3):
        tr.step(x, ht)
    torch.cuda.synchronize()
    chk = torch.stack([p.detach().double().sum() for p in m.parameters()]).sum().reshape(1).cpu()
    both = [torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)]
    dist.all_gather(both, chk)
    q.put((rank, worst, emet, float(both[0]), float(both[1])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_reproduce_the_global_batch():
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, worst, emet, c0, c1 in res:
        assert worst < 2e-5, (rank, worst)                      # mean of the ranks' gradients == gradients of the global batch
        assert emet < 1e-6, (rank, emet)                        # the logged scalars, rank-mean == global-batch values
        assert c0 == c1, (rank, c0, c1)                         # replicas identical after three Adam steps


def _worker_other(rank, world, port, q, which):
    """WaveFlow (autograd path + GradSync.all_reduce_params: one collective behind the backward) and WSRGlow (FlowTrainer with the
    embedding-table bucket) with two ranks: BASELINE.json configs[3] / [4] train data-parallel like configs[2] (train.py:51-53,73-78)."""
    import constant_memory_waveglow_amd as cm
    from constant_memory_waveglow_amd.parallel import FlowTrainer, GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if which == "waveflow":
        name = "wf8"
        cfg = fill.WF_CONFIGS[name]
        _, N, F = fill.WF_SHAPES[name]
        specs = fill.waveflow_param_specs(cfg)
        P = fill.fill_params(specs, name + "/")

        def model():
            m = cm.WaveFlow(memory_efficient=False, bias=False, **dict({"use_conv1x1": False}, **cfg))
            m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
            return m.to(dev)
        audio, h = fill.waveflow_inputs("wfdp2", 4, N, F, cfg["n_mels"])
        m, ref = model(), model()
        if rank == 1:
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(0.01)
        sync = GradSync()
        params = list(m.parameters())
        sync.broadcast_params(params)
        for p, r in zip(m.parameters(), ref.parameters()):
            assert torch.equal(p, r)
        mine = slice(2 * rank, 2 * rank + 2)
        crit = cm.WaveGlowLoss(fill.SIGMA)
        z, ld = m(torch.from_numpy(audio[mine]).to(dev), torch.from_numpy(h[mine]).to(dev))
        crit(z, ld).backward()
        sync.all_reduce_params(params)
        zf, ldf = ref(torch.from_numpy(audio).to(dev), torch.from_numpy(h).to(dev))
        crit(zf, ldf).backward()
        worst = 0.0
        for (n, p), r in zip(m.named_parameters(), ref.parameters()):
            if n.endswith("start.weight_v"):                  # exact gradient zero (w = g sign(v)): rounding noise on both sides
                continue
            worst = max(worst, float((p.grad - r.grad).abs().max() / r.grad.abs().max().clamp_min(1e-30)))
    else:
        name = "wsr"                                            # the shipped width (229.7 M parameters: a 919 MB gradient exchange in 14 buckets)
        cfg = fill.CONFIGS[name]
        specs = fill.model_param_specs(cfg)
        P = fill.fill_params(specs, name + "/")
        P.update(fill.wsr_tables(name + "/"))

        def model():
            m = cm.WSRGlow(upsample_rate=fill.WSR_RATE[name], memory_efficient=True, bias=False, **fill.WSR_KW)
            sd = {k: torch.from_numpy(v) for k, v in P.items()}
            sd["window"] = torch.hann_window(16)
            m.load_state_dict(sd)
            return m.to(dev)
        rate = fill.WSR_RATE[name]
        N = fill.SHAPES[name][1]
        audio = fill.uniform("wsrdp2/x", (4, N), -1.0, 1.0)
        c = fill.uniform("wsrdp2/c", (4, N // rate), -0.9, 0.9)
        m, ref = model(), model()
        tr = FlowTrainer(m, 1.0)
        assert tr.sync.world == 2 and len(tr.frontend) == 2
        mine = slice(2 * rank, 2 * rank + 2)
        tr.step(torch.from_numpy(audio[mine]).to(dev), torch.from_numpy(c[mine]).to(dev))
        trf = FlowTrainer(ref, 1.0)
        trf.sync.skip = True                                    # the same global batch in one process, no collective
        trf.step(torch.from_numpy(audio).to(dev), torch.from_numpy(c).to(dev))
        worst = 0.0
        for b in range(tr.n_buckets):
            s, e = tr.fg.bucket_ranges[b]
            if e > s:
                worst = max(worst, float((tr.fg.flat[s:e] - trf.fg.flat[s:e]).abs().max() / trf.fg.flat[s:e].abs().max().clamp_min(1e-30)))
    torch.cuda.synchronize()
    q.put((rank, worst))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("which", ["waveflow", "wsrglow"])
def test_two_ranks_other_models(which):
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_other, args=(r, 2, port, q, which)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst < 5e-5, (which, rank, worst)               # mean of the ranks' gradients == gradients of the global batch
