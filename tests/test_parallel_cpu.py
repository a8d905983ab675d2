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
    assert waveglow_buckets(12, 8, extra=2)[-2:] == [13, 13]                     # WSRGlow's embedding tables: their own bucket
    # the metric tail rides in the LAST bucket's collective and in no parameter range
    fg = FlatGrads(params, [1, 0, 1, 2], tail=4)
    assert fg.bucket_ranges == [(0, 4), (4, 12), (12, 13)] and fg.tail_off == 16 and fg.flat.numel() == 20 and fg.tail.numel() == 4
    assert fg.comm_slice(0).numel() == 4 and fg.comm_slice(2).numel() == 20 - 12 and fg.bucket(2).numel() == 1
    fg.tail.fill_(3.0)
    assert float(fg.comm_slice(2).sum()) == 12.0 and float(fg.bucket(2).sum()) == 0.0


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
    fg = FlatGrads(params, waveglow_buckets(cfg["flows"], cfg["depth"]), tail=4)
    for v, g in zip(fg.views, r["grads"]):
        v.copy_(torch.from_numpy(g))
    # the four scalars the reference logs with sync_dist=True (model/lightning.py:58-64), this rank's values, in the tail
    zt, ldt = torch.from_numpy(r["z"]), torch.from_numpy(r["logdet"])
    mine_metrics = torch.tensor([float(ldt.sum() / zt.numel()), float(zt.mean()), float(zt.std()), r["loss"]])
    fg.tail.copy_(mine_metrics)
    sync = GradSync()
    # after_bucket is where FlatAdam hangs the optimizer step: it must see each bucket already averaged, in the order given
    seen = []
    half = [b.clone() for b in (fg.bucket(i) for i in range(len(fg.bucket_ranges)))]
    order = list(range(cfg["flows"] - 1, -1, -1)) + [cfg["flows"]]
    sync.all_reduce(fg, order=order, after_bucket=lambda b: seen.append((b, float((fg.bucket(b) - half[b]).abs().max()))))
    assert [b for b, _ in seen] == order, seen
    assert any(d > 0 for _, d in seen)                                  # the buckets had been reduced when the callback ran
    # the 4-float metric vector came out as the mean over the ranks (what Lightning's sync_dist does with each logged scalar)
    both = [torch.zeros(4), torch.zeros(4)]
    dist.all_gather(both, mine_metrics)
    assert torch.allclose(fg.tail, (both[0] + both[1]) / 2, rtol=1e-6, atol=1e-7), (fg.tail, both)
    # autograd-trained models (WSRGlow, WaveFlow): one coalesced mean all-reduce of p.grad
    ps = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(1))]
    ps[0].grad = torch.full((5,), float(rank + 1))
    ps[1].grad = torch.arange(6, dtype=torch.float32).view(2, 3) * (rank + 1)
    sync.all_reduce_params(ps)
    assert torch.allclose(ps[0].grad, torch.full((5,), 1.5)) and torch.allclose(ps[1].grad, torch.arange(6, dtype=torch.float32).view(2, 3) * 1.5)
    assert ps[2].grad is None
    # GradSync.skip (bench.py's `exposed_ms` run: the same step WITHOUT its collectives, every rank alike): nothing is exchanged, the
    # per-bucket callbacks still run in order
    sync.skip = True
    ps[0].grad = torch.full((5,), float(rank + 1))
    sync.all_reduce_params(ps)
    assert torch.equal(ps[0].grad, torch.full((5,), float(rank + 1)))
    before, seen2 = fg.flat.clone(), []
    sync.all_reduce(fg, order=order, after_bucket=lambda b: seen2.append(b))
    assert seen2 == order and torch.equal(fg.flat, before)
    sync.skip = False
    # replicas start identical
    probe = [torch.full((3,), float(rank)), None, torch.nn.Parameter(torch.full((2, 4), 7.0 + rank)), torch.full((1,), 5.0 * rank)]
    sync.broadcast_params(probe)                                      # ONE flat collective over all of them, copied back per tensor
    assert float(probe[2].detach().sum()) == 56.0 and float(probe[3][0]) == 0.0 and probe[2].shape == (2, 4)
    if rank == 0:
        full = orc.train_step(oc, tab, audio, h, fill.SIGMA)
        worst = max(float(np.abs(v.numpy() - g).max() / max(np.abs(g).max(), 1e-30)) for v, g in zip(fg.views, full["grads"]))
        # equal per-rank batches: the rank-mean of logdet/numel, z.mean and the loss IS the global-batch value
        zf, lf = torch.from_numpy(full["z"]), torch.from_numpy(full["logdet"])
        assert abs(float(fg.tail[0]) - float(lf.sum() / zf.numel())) < 1e-6 and abs(float(fg.tail[1]) - float(zf.mean())) < 1e-6
        assert abs(float(fg.tail[3]) - full["loss"]) < 1e-6
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
