"""Data-parallel path on CPU: 2 gloo ranks.  The gradient plumbing (flat buffer, per-flow buckets, mean all-reduce)
is the product code; the per-rank gradients are produced by the CPU oracle here because the HIP kernels need a GPU.
Checks the semantic the reference relies on with DDP: mean all-reduce of per-rank mean-loss grads
== gradients of the global batch, including the batch-shared logdet(W) term."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fill
from constant_memory_waveglow_amd.parallel import FlatGrads, GradSync, waveglow_buckets


def test_flat_grads_layout():
    params = [torch.zeros(3), torch.zeros(2, 2), torch.zeros(5), torch.zeros(1)]
    fg = FlatGrads(params, [1, 0, 1, 2])
    assert fg.total == 13
    assert fg.bucket_ranges == [(0, 4), (4, 12), (12, 13)]
    fg.views[1].fill_(7.0)
    assert float(fg.bucket(0).sum()) == 28.0 and float(fg.bucket(1).sum()) == 0.0
    assert [v.shape for v in fg.views] == [p.shape for p in params]
    ids = waveglow_buckets(12, 8)
    assert len(ids) == 459 and ids[:3] == [12] * 3 and ids[3:15] == list(range(12)) and ids[15] == 0 and ids[-1] == 11


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import wg_oracle as orc
    name = "micro"
    cfg = fill.CONFIGS[name]
    _, N, F = fill.SHAPES[name]
    specs = fill.model_param_specs(cfg)
    tab = fill.table(specs, fill.fill_params(specs, name + "/"))
    audio, h = fill.inputs("dp", 4, N, F, cfg["n_mels"])            # global batch 4, 2 per rank
    oc = orc.make_config(**cfg)
    mine = slice(2 * rank, 2 * rank + 2)
    r = orc.train_step(oc, tab, audio[mine], h[mine], fill.SIGMA)
    params = [torch.from_numpy(p) for p in tab]
    fg = FlatGrads(params, waveglow_buckets(cfg["flows"], cfg["depth"]))
    for v, g in zip(fg.views, r["grads"]):
        v.copy_(torch.from_numpy(g))
    sync = GradSync()
    # after_bucket is where FlatAdam hangs the optimizer step: it must see each bucket already averaged, in the order given
    seen = []
    half = [b.clone() for b in (fg.bucket(i) for i in range(len(fg.bucket_ranges)))]
    order = list(range(cfg["flows"] - 1, -1, -1)) + [cfg["flows"]]
    sync.all_reduce(fg, order=order, after_bucket=lambda b: seen.append((b, float((fg.bucket(b) - half[b]).abs().max()))))
    assert [b for b, _ in seen] == order, seen
    assert any(d > 0 for _, d in seen)                                  # the buckets had been reduced when the callback ran
    # autograd-trained models (WSRGlow, WaveFlow): one coalesced mean all-reduce of p.grad
    ps = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(1))]
    ps[0].grad = torch.full((5,), float(rank + 1))
    ps[1].grad = torch.arange(6, dtype=torch.float32).view(2, 3) * (rank + 1)
    sync.all_reduce_params(ps)
    assert torch.allclose(ps[0].grad, torch.full((5,), 1.5)) and torch.allclose(ps[1].grad, torch.arange(6, dtype=torch.float32).view(2, 3) * 1.5)
    assert ps[2].grad is None
    # replicas start identical
    probe = [torch.full((3,), float(rank))]
    sync.broadcast_params(probe)
    if rank == 0:
        full = orc.train_step(oc, tab, audio, h, fill.SIGMA)
        worst = max(float(np.abs(v.numpy() - g).max() / max(np.abs(g).max(), 1e-30)) for v, g in zip(fg.views, full["grads"]))
        q.put((worst, float(probe[0][0])))
    else:
        q.put((None, float(probe[0][0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_mean_allreduce_equals_global_batch_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    worst = [w for w, _ in res if w is not None][0]
    assert worst < 2e-5, worst
    assert all(v == 0.0 for _, v in res)
