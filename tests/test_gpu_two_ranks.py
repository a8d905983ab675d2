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
