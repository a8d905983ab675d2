"""Data-parallel training of the flow: one process per GPU, gradients averaged with RCCL all-reduce over xGMI.

Replaces what the reference gets from Lightning's DDPPlugin (train.py:51-53,73-78): per-GPU batch = global // world,
mean-all-reduce of every parameter gradient each step, replicas start from rank 0's weights.  The path has no
other exchange step: batch items are independent units.

Gradients of one step live in ONE flat fp32 buffer (FlatGrads) carved in parameter-table order, so the all-reduce
runs on contiguous slices: one bucket per flow, issued in backward order (last flow first), plus the upsampler /
1x1 bucket.  xGMI is point to point, so a few large (>=8 MB) buckets are preferred over DDP's 25 MB default mix.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


class FlatGrads:
    """A flat gradient buffer with per-parameter views and per-bucket slices.  `tail` extra floats follow the last bucket: they
    travel in that bucket's collective (the four logged training scalars ride there) but are not part of any parameter range."""

    def __init__(self, params: Sequence[torch.Tensor], bucket_of: Sequence[int], tail: int = 0):
        assert len(params) == len(bucket_of)
        self.sizes = [p.numel() for p in params]
        order = sorted(range(len(params)), key=lambda i: (bucket_of[i], i))
        self.offsets = [0] * len(params)
        self.bucket_ranges: List[Tuple[int, int]] = []
        off = 0
        cur, start = None, 0
        for i in order:
            if bucket_of[i] != cur:
                if cur is not None:
                    self.bucket_ranges.append((start, off))
                cur, start = bucket_of[i], off
            self.offsets[i] = off
            off += self.sizes[i]
        if cur is not None:
            self.bucket_ranges.append((start, off))
        self.total = off
        self.tail_off = (off + 3) // 4 * 4                     # 16-byte aligned
        self.tail_len = int(tail)
        self.flat = torch.zeros(self.tail_off + self.tail_len, dtype=params[0].dtype, device=params[0].device)
        self.views = [self.flat[o:o + n].view_as(p) for o, n, p in zip(self.offsets, self.sizes, params)]
        self.tail = self.flat[self.tail_off:self.tail_off + self.tail_len]

    def bucket(self, b: int) -> torch.Tensor:
        """the parameter gradients of bucket b"""
        s, e = self.bucket_ranges[b]
        return self.flat[s:e]

    def comm_slice(self, b: int) -> torch.Tensor:
        """what bucket b's collective moves: its gradients, plus the tail for the last bucket"""
        s, e = self.bucket_ranges[b]
        if b == len(self.bucket_ranges) - 1 and self.tail_len:
            e = self.tail_off + self.tail_len
        return self.flat[s:e]


def waveglow_buckets(n_flows: int, depth: int, extra: int = 0, bias: bool = False) -> List[int]:
    """bucket id per parameter-table entry: flow k (its WN and its 1x1 weight) -> bucket k; upsampler -> bucket n_flows;
    `extra` trailing entries (WSRGlow's two embedding tables) -> bucket n_flows + 1."""
    ids = [n_flows] * 3 + list(range(n_flows))
    for k in range(n_flows):
        ids += [k] * (4 + 4 * depth + 1 + (2 + 2 * depth + 1 if bias else 0))
    return ids + [n_flows + 1] * extra


class GradSync:
    """Mean all-reduce of a FlatGrads over the process group, bucket by bucket in backward order."""

    def __init__(self, process_group=None, force_collectives=None):
        """force_collectives: run the collectives (side stream, events, RCCL calls) even in a group of ONE rank -- how the
        path is exercised on a single GPU (tests, WG_BENCH_FORCE_DIST=1); default: the environment variable decides."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self._avg = backend == "nccl"          # RCCL has a native AVG; gloo does not
        self._comm = None
        import os
        if force_collectives is None:
            force_collectives = os.environ.get("WG_BENCH_FORCE_DIST") == "1"
        self._force = dist.is_initialized() and bool(force_collectives)
        self.skip = False                      # diagnostics (bench.py's `exposed_ms`): run the step WITHOUT its collectives, all ranks alike

    def all_reduce(self, fg: FlatGrads, order: Sequence[int] = None, events=None, after_bucket=None):
        """events[b] (torch.cuda.Event, optional): bucket b's gradients are final once the event has fired; its all-reduce is
        enqueued behind that event on a side stream, so it overlaps whatever backward work is still queued on the main stream.
        after_bucket(b) (optional) runs right behind bucket b's reduction -- on the side stream when there is one -- which is
        where FlatAdam puts the optimizer step of that bucket."""
        order = list(order) if order is not None else list(range(len(fg.bucket_ranges) - 1, -1, -1))
        if (self.world == 1 and not self._force) or self.skip:
            if after_bucket is not None:
                for b in order:
                    after_bucket(b)
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        works = []
        if events is not None and fg.flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=fg.flat.device)
            with torch.cuda.stream(self._comm):
                for b in order:
                    t = fg.comm_slice(b)
                    if t.numel() == 0:
                        continue
                    self._comm.wait_event(events[b])
                    w = dist.all_reduce(t, op=op, group=self.pg, async_op=True)
                    if after_bucket is not None:
                        w.wait()                                  # the SIDE stream waits for the collective ...
                        if not self._avg:
                            t.div_(self.world)
                        after_bucket(b)                           # ... and runs the bucket's optimizer step behind it
                    else:
                        works.append((w, t))
            if after_bucket is not None:
                torch.cuda.current_stream(fg.flat.device).wait_stream(self._comm)
                return
        else:
            for b in order:
                t = fg.comm_slice(b)
                if t.numel() == 0:
                    continue
                works.append((dist.all_reduce(t, op=op, group=self.pg, async_op=True), t))
        for i, (w, t) in enumerate(works):
            w.wait()                      # the current (main) stream waits for the collective
            if not self._avg:
                t.div_(self.world)
        if after_bucket is not None:
            for b in order:
                if fg.bucket(b).numel():
                    after_bucket(b)

    def all_reduce_params(self, params: Sequence[torch.Tensor]):
        """Mean all-reduce of `p.grad` for models trained through autograd (WaveFlow): the gradients are copied into one flat buffer,
        reduced with ONE collective and copied back.  Not bucketed and not overlapped, on purpose: WaveFlow has 5.95 M parameters =
        24 MB, i.e. 2 * 7/8 * 24 MB = 42 MB per GPU through >= 153 GB/s of xGMI = about 0.3 ms behind a 51 ms step (0.6 %), and its
        autograd node (`_WaveFlowFn`) hands over all gradients at once when the backward call returns, so there is no earlier point
        at which a bucket would be final.  (WaveGlow / WSRGlow, 215 / 919 MB, go through FlowTrainer's per-flow buckets.)"""
        if (self.world == 1 and not self._force) or self.skip:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.pg)
        if not self._avg:
            flat.div_(self.world)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def broadcast_params(self, params: Sequence[torch.Tensor], src: int = 0):
        """src: the GLOBAL rank whose weights every replica takes (it must belong to the group).  Replicas start identical (what DDP does when it wraps the module): ONE broadcast of all parameters as a flat buffer
        (WaveGlow-256ch: 459 tensors, 214.6 MB -- 459 separate collectives would each pay RCCL's launch latency), copied back into
        the parameters on the receiving ranks."""
        if self.world == 1 and not self._force:
            return
        params = [p for p in params if p is not None]
        if not params:
            return
        if len({(p.dtype, p.device) for p in params}) != 1:
            raise ValueError("broadcast_params: the parameters must share one dtype and one device (torch.cat would promote silently)")
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in params])
            dist.broadcast(flat, src=src, group=self.pg)          # `src` is a GLOBAL rank (torch.distributed's convention)
            if dist.get_rank() != src or self._force:             # ... so it is compared with this process's global rank
                off = 0
                for p in params:
                    n = p.numel()
                    p.detach().copy_(flat[off:off + n].view_as(p))
                    off += n


METRIC_NAMES = ("logdet", "z_mean", "z_std", "loss")      # the keys LightModel.training_step logs (model/lightning.py:58-64)


class FlowTrainer:
    """The training step of model/lightning.py:52-65 without Lightning:
        z, logdet = model(x, h); loss = NLL(z, logdet); backward to every parameter gradient; all-reduce(mean);
        the logged scalars logdet.sum()/z.numel(), z.mean(), z.std(), loss, mean-reduced over the ranks (sync_dist=True).
    Runs the HIP engine directly (no autograd graph) and leaves the gradients in p.grad (views of one flat buffer).
    `model` is a WaveGlow or a WSRGlow (a WaveGlow whose conditioning is built from the low-rate signal by `_get_cond`:
    step(x, c) then takes that signal, and the two embedding tables get their own gradient bucket behind the upsampler's)."""

    def __init__(self, model, sigma: float, elementwise_mean: bool = True, process_group=None, repack_every_step: bool = True,
                 force_collectives=None):
        """repack_every_step: the reference recomputes w = g v / ||v|| in a forward-pre-hook on every call (utils.py:14-16), and in
        training the parameters change between steps anyway.  The engine caches its packed weights by parameter version; a step
        timed on frozen parameters would silently skip that work, so the trainer re-packs on every step unless told otherwise.
        force_collectives: see GradSync."""
        from . import engine
        self.repack_every_step = repack_every_step
        self._engine_mod = engine
        self.model = model
        self.sigma, self.mean = sigma, elementwise_mean
        self.sync = GradSync(process_group, force_collectives)
        self.frontend = [model.mu_enc[1].weight, model.angle_embed.embed.weight] if hasattr(model, "_get_cond") else []
        self.flow_table = [t for t in model.param_table()]              # the C-ABI table of the flow (wg_train_step)
        self.table = self.flow_table + self.frontend
        wn0 = model.WNs[0].F
        ids = waveglow_buckets(len(model.WNs), len(wn0.layers), extra=len(self.frontend), bias=getattr(wn0, "has_bias", False))
        live = [(t, b) for t, b in zip(self.table, ids) if t is not None]
        self.fg = FlatGrads([t for t, _ in live], [b for _, b in live], tail=len(METRIC_NAMES))
        it = iter(self.fg.views)
        self.grad_views = [next(it) if t is not None else None for t in self.table]
        self.sync.broadcast_params([t for t in self.table if t is not None])
        self.n_flows = len(model.WNs)
        self.n_buckets = len(self.fg.bucket_ranges)
        self.metrics = self.fg.tail       # [logdet.sum()/z.numel(), z.mean(), z.std(), loss] of the last step, reduced over the ranks
        self.optimizer = None             # a FlatAdam attaches itself here; step() then updates the weights as well
        self.want_dh, self.last_dh = False, None      # want_dh: step() also keeps d loss / d h (wg_train_step's dh output) in last_dh
        self.events = None
        if self.table[0].is_cuda and (self.sync.world > 1 or self.sync._force):
            with torch.cuda.device(self.table[0].device):
                self.events = [torch.cuda.Event() for _ in range(self.n_buckets)]
                for e in self.events:
                    e.record()            # materialise the underlying hipEvent_t so its handle can cross the C ABI

    @torch.no_grad()
    def step(self, x: torch.Tensor, h: torch.Tensor):
        """x [B, N] audio; h: the conditioning [B, n_mels, frames] (WaveGlow) or the low-rate signal [B, N / rate] (WSRGlow, which
        clips it in place like the reference, wsrglow.py:38).  Returns this rank's (loss, z, logdet)."""
        eng, E = self.model._engine, self._engine_mod
        if self.repack_every_step:
            eng.packed.key = None             # a training step always re-materialises the weight-normed weights (see __init__)
        nflow = len(self.flow_table)
        table = [None if t is None else t.detach() for t in self.flow_table]
        need = [t is not None and t.requires_grad for t in self.flow_table]
        c = None
        if self.frontend:
            c = h.clamp_(-1.0, 1.0)
            h = E.wsr_cond(c, self.frontend[0].detach(), self.frontend[1].detach())
        loss, z, logdet, _, dh = eng.train_step(table, x, h, self.sigma, self.mean, need, grads_out=self.grad_views[:nflow],
                                                need_dh=bool(self.frontend) or self.want_dh,
                                                flow_events=self.events[:self.n_flows + 1] if self.events else None,
                                                keep=not self.model.mem_efficient, metrics=self.metrics)   # one C call: wg_train_step
        # buckets become final in the order backward retires the flows: last flow first (first flow first in reverse_mode), then the
        # upsampler, then (WSRGlow) the embedding tables, whose gradients come out of the conditioning gradient
        flows = range(self.n_flows) if self.model._reverse_mode else range(self.n_flows - 1, -1, -1)
        order = list(flows) + [self.n_flows]
        self.last_dh = dh if self.want_dh else None      # d loss / d conditioning of this step (want_dh: off by default, nothing trains on it)
        if self.frontend:
            E.wsr_cond_backward(c, dh, out=(self.grad_views[nflow], self.grad_views[nflow + 1]))
            del dh
            if self.events:
                self.events[self.n_flows + 1].record(torch.cuda.current_stream(x.device))
            order.append(self.n_flows + 1)
        opt = self.optimizer
        self.sync.all_reduce(self.fg, order=order, events=self.events,
                             after_bucket=opt.step_bucket if opt is not None else None)
        if opt is not None:
            opt.finish_step()
        for t, g in zip(self.table, self.grad_views):
            if t is not None:
                t.grad = g
        return loss, z, logdet

    def metrics_dict(self):
        """{'logdet', 'z_mean', 'z_std', 'loss'} of the last step as Python floats (one device->host copy)."""
        return dict(zip(METRIC_NAMES, self.metrics.tolist()))


class FlatAdam:
    """torch.optim.Adam for a FlowTrainer, run by `wg_adam_step` on flat buffers.

    The reference builds `torch.optim.Adam(self.parameters(), **optimizer.args)` (model/lightning.py:41-44; lr 1e-4 in
    configs/waveglow_LJ_speech.json).  Here the parameters of the trainer's table are re-pointed to views of ONE flat fp32
    buffer laid out like the gradient buffer (one bucket per flow), so the update of a bucket is a single HBM-bound launch over a
    contiguous range, issued right behind that bucket's gradient all-reduce on the communication stream: the optimizer step of the
    last flows overlaps the backward of the first ones.  `state_dict()` / `load_state_dict()` use torch's per-parameter layout
    (`exp_avg`, `exp_avg_sq`, `step`), so optimizer checkpoints interchange with torch.optim.Adam.
    """

    def __init__(self, trainer: "FlowTrainer", lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        from . import _lib
        self._lib = _lib
        self.trainer = trainer
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        fg = trainer.fg
        live = [t for t in trainer.table if t is not None]
        if not live[0].is_cuda:
            raise _lib.WgError("FlatAdam runs on the HIP device only (no CPU fallback)")
        self.flat = torch.zeros_like(fg.flat)                # (zeros: the alignment padding and the metric tail are not parameters)
        for t, o, n in zip(live, fg.offsets, fg.sizes):
            view = self.flat[o:o + n].view_as(t)
            view.copy_(t.data)
            t.data = view                                   # the module's parameters now live in the flat buffer
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self._live = live
        trainer.optimizer = self

    def step_bucket(self, b: int):
        """Adam on bucket b (called by GradSync behind the bucket's all-reduce, on whatever stream is current there)."""
        import ctypes as C
        s, e = self.trainer.fg.bucket_ranges[b]
        if e == s:
            return
        off = 4 * s
        L = self._lib.lib()
        st = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
        self._lib.check(L.wg_adam_step(C.c_void_p(self.flat.data_ptr() + off), C.c_void_p(self.trainer.fg.flat.data_ptr() + off),
                                       C.c_void_p(self.exp_avg.data_ptr() + off), C.c_void_p(self.exp_avg_sq.data_ptr() + off),
                                       C.c_size_t(e - s), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                       self.step_count + 1, st), "wg_adam_step")

    def finish_step(self):
        self.step_count += 1
        self.trainer.model._engine.packed.key = None        # weights changed behind torch's version counters: re-pack next step

    def step(self):
        """stand-alone use (gradients already reduced): all buckets on the current stream"""
        for b in range(len(self.trainer.fg.bucket_ranges)):
            self.step_bucket(b)
        self.finish_step()

    # ---- checkpoints: torch.optim.Adam's layout --------------------------------------------------------------------------
    def state_dict(self):
        fg = self.trainer.fg
        state = {}
        for i, (t, o, n) in enumerate(zip(self._live, fg.offsets, fg.sizes)):
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[o:o + n].view_as(t).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(t).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "params": list(range(len(self._live)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        fg = self.trainer.fg
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"]), float(g["weight_decay"])
        steps = set()
        for i, (t, o, n) in enumerate(zip(self._live, fg.offsets, fg.sizes)):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.exp_avg[o:o + n].view_as(t).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + n].view_as(t).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise self._lib.WgError("FlatAdam keeps one step counter; the checkpoint has several: %s" % sorted(steps))
        self.step_count = steps.pop() if steps else 0


def load_reference_checkpoint(model, checkpoint, prefix="model."):
    """Loads the flow's weights from a checkpoint written by the reference's LightModel (pytorch-lightning layout:
    checkpoint["state_dict"] with the flow under `model.` next to the conditioner's buffers -- model/lightning.py:38-40,
    inference.py:17).  Accepts the checkpoint dict or a bare state dict; returns load_state_dict's result."""
    sd = checkpoint.get("state_dict", checkpoint)
    own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if any(k.startswith(prefix) for k in sd) else dict(sd)
    return model.load_state_dict(own)
