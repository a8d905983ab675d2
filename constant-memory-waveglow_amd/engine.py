"""Thin torch-side plumbing over the C ABI: device buffers, streams, caches.  No arithmetic happens here.

Everything that computes is a HIP kernel behind include/wgflow.h.  PyTorch only owns the device memory
(tensors as workspaces / packed-weight buffers) and the stream the kernels are enqueued on.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import WgConfig, WgWnDims, WgError, check


_CAPTURE_LOCK = __import__("threading").Lock()


def _graph_mode():
    """How synthesis calls use hipGraphs (WG_GRAPHS): "0" never, "1" always (captured on the first call of a shape), unset = auto: a small
    call (at most 65 536 samples: ~250 launches of a few microseconds each) whose exact shape and buffers have been seen twice before is
    captured on its third occurrence and replayed from then on.  Measured on MI355X (gpurun_out/r04f_stdout.txt): one 0.7 s utterance
    between two synchronisations 2.72 -> 2.62 ms, host enqueue time 0.97 -> 0.26 ms; back to back the device is the bound either way
    (2.47 / 2.44 ms).  Variable-length serving never repeats a shape three times in a row of cache entries and stays on direct launches."""
    import os
    v = os.environ.get("WG_GRAPHS", "auto")
    return v if v in ("0", "1") else "auto"


def _stream(device=None):
    """the stream the kernels are enqueued on: torch's current stream OF THE TENSORS' DEVICE (not of the current device)"""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(fn):
    """Runs an engine entry point with the device of its first tensor argument current: the library sizes its launches with
    hipGetDevice and torch's allocations / streams follow the current device, so a model on cuda:1 must not be driven from
    cuda:0's context."""
    import functools

    def first_tensor(xs):
        for a in xs:
            if torch.is_tensor(a):
                return a
            if isinstance(a, (list, tuple)):
                t = first_tensor(a)
                if t is not None:
                    return t
        return None

    @functools.wraps(fn)
    def run(*args, **kwargs):
        t = first_tensor(args) if args else None
        if t is None:
            t = first_tensor(list(kwargs.values()))
        if t is None or not t.is_cuda:
            return fn(*args, **kwargs)                  # require_device raises the proper error
        with torch.cuda.device(t.device):
            return fn(*args, **kwargs)
    return run


def require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise WgError("constant-memory-waveglow_amd runs on an MI355X (HIP) device only: got a %s tensor. "
                          "There is no CPU fallback." % t.device)
        if t.dtype != torch.float32:
            raise WgError("the HIP engine computes in float32 (got %s)" % t.dtype)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise WgError("tensors of one call sit on different devices (%s and %s)" % (dev, t.device))


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _table(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def make_config(flows, n_group, n_early_every, n_early_size, hop_size, n_mels,
                residual_channels, dilation_channels, skip_channels, depth, radix, precision=None, reverse_mode=False, bias=False):
    up = hop_size // n_group                  # reference model/waveglow.py:125
    k = up * 2 + 1                            # :126
    pad = k // 2 - up // 2                    # :128-129
    return WgConfig(flows, n_group, n_early_every, n_early_size, n_mels, up, k, pad,
                    residual_channels, dilation_channels, skip_channels, depth, radix,
                    _lib.default_precision() if precision is None else precision, int(bool(reverse_mode)), 0, int(bool(bias)))


class _Buffers:
    """Zero-initialised device workspaces keyed by (device, tag, shape...).  The kernels keep the zero halo of
    every activation plane intact, so a workspace is zeroed once when it is allocated.  At most `keep` shapes stay
    resident (least recently used goes first), so variable-length inference does not pile up multi-GB buffers."""

    def __init__(self, keep=4):
        self._ws = {}
        self._keep = keep

    def get(self, key, nbytes, device):
        buf = self._ws.pop(key, None)
        if buf is None or buf.numel() < nbytes or buf.device != device:
            if nbytes == 0:
                raise WgError("configuration/shape rejected by the HIP engine (workspace query returned 0)")
            while len(self._ws) >= self._keep:
                self._ws.pop(next(iter(self._ws)))
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self._ws[key] = buf                      # (re)insert as most recently used
        return buf

    def clear(self):
        self._ws.clear()


class PackedWeights:
    """Materialised (weight-normed, k-major) weights; re-packed only when a parameter changed."""

    def __init__(self):
        self.buf = None
        self.key = None
        self._pending = None

    def stale(self, params):
        """True: `params` differ from what `buf` holds.  The buffer then counts as EMPTY until commit(): a pack that raises half way
        (a CPU or half parameter further down the table, an unsupported configuration, a failed launch) must not leave a key behind
        under which a retry would run on the old or half-written weights."""
        key = tuple((p.data_ptr(), p._version) if p is not None else None for p in params)
        if key != self.key or self.buf is None:
            self.key, self._pending = None, key
            return True
        return False

    def commit(self):
        """the pack behind the last stale() succeeded"""
        self.key = self._pending


class ModelEngine:
    """Whole-model entry points (wg_pack_weights / wg_forward / wg_inverse / wg_backward)."""

    def __init__(self, cfg: WgConfig):
        self.cfg = cfg
        self.cfg_keep = WgConfig.from_buffer_copy(cfg)      # the same model in stored-activation mode (memory_efficient=False)
        self.cfg_keep.keep_activations = 1
        self._kept_gen = 0                                  # bumped by every call that rewrites a stored-activation workspace
        self.buffers = _Buffers()
        self.packed = PackedWeights()
        self.n_params = None
        self._graphs = {}                        # (device, shape, frames, packed buffer, workspace) -> captured inverse (hipGraph)
        self._graph_seen = {}                    # auto mode: how often a small call's key has occurred

    def _pack(self, params, device):
        L = _lib.lib()
        if self.n_params is None:
            self.n_params = L.wg_param_count(C.byref(self.cfg))
        if len(params) != self.n_params:
            raise WgError("parameter table has %d entries, expected %d" % (len(params), self.n_params))
        if self.packed.stale(params) or self.packed.buf.device != device:
            require_device(*params)
            nbytes = L.wg_packed_bytes(C.byref(self.cfg))
            if nbytes == 0:
                raise WgError("WaveGlow configuration not supported by the HIP kernels "
                              "(channels must be multiples of 32, odd radix <= 9, n_group <= 32)")
            if self.packed.buf is None or self.packed.buf.numel() < nbytes or self.packed.buf.device != device:
                self.packed.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            check(L.wg_pack_weights(C.byref(self.cfg), _table(params), _p(self.packed.buf), _stream()), "wg_pack_weights")
            self.packed.commit()
        return self.packed.buf

    def _ws(self, B, N, mode, device):
        nbytes = _lib.lib().wg_workspace_bytes(C.byref(self.cfg), B, N, mode)
        return self.buffers.get((device, mode, B, N), nbytes, device)

    def _ws_keep(self, B, N, device):
        nbytes = _lib.lib().wg_workspace_bytes(C.byref(self.cfg_keep), B, N, 1)
        return self.buffers.get((device, "keep", B, N), nbytes, device)

    def _launch(self, pk, ws, x, h, inverse, cfg=None):
        B, N = x.shape
        out = torch.empty_like(x)
        logdet = torch.empty(B, dtype=torch.float32, device=x.device)
        fn = _lib.lib().wg_inverse if inverse else _lib.lib().wg_forward
        check(fn(C.byref(cfg or self.cfg), _p(pk), _p(x), _p(h), B, N, h.shape[2], _p(out), _p(logdet), _p(ws), ws.numel(), _stream()),
              "wg_inverse" if inverse else "wg_forward")
        return out, logdet

    @on_device
    def run_keep(self, params, x, h):
        """wg_forward in stored-activation mode.  Returns (z, logdet, kept): `kept` = (generation, workspace) names the
        activations this call left behind; `backward(kept=...)` uses them if no later call overwrote them."""
        require_device(x, h, *params)
        x, h = x.contiguous(), h.contiguous()
        B, N = x.shape
        pk = self._pack(params, x.device)
        ws = self._ws_keep(B, N, x.device)
        self._kept_gen += 1
        z, logdet = self._launch(pk, ws, x, h, False, self.cfg_keep)
        return z, logdet, (self._kept_gen, ws)

    @on_device
    def run(self, params, x, h, inverse):
        require_device(x, h, params[0])            # (the whole table is checked when it is packed: _pack)
        x, h = x.contiguous(), h.contiguous()
        B, N = x.shape
        pk = self._pack(params, x.device)
        ws = self._ws(B, N, 0, x.device)
        mode = _graph_mode() if inverse else "0"
        if mode != "0" and not torch.cuda.is_current_stream_capturing():
            key = (x.device, tuple(x.shape), h.shape[2], pk.data_ptr(), ws.data_ptr())
            if mode == "1" or key in self._graphs:
                return self._replay_inverse(key, pk, ws, x, h)
            if x.numel() <= 65536:                          # auto: capture a small call on its third occurrence
                n = self._graph_seen.get(key, 0) + 1
                if len(self._graph_seen) > 16:
                    self._graph_seen.clear()
                self._graph_seen[key] = n
                if n >= 3:
                    try:
                        return self._replay_inverse(key, pk, ws, x, h)
                    except Exception:                       # noqa: BLE001 -- a capture that fails only costs the graph: direct launches go on
                        self._graph_seen[key] = -(1 << 30)
        return self._launch(pk, ws, x, h, inverse)

    def _replay_inverse(self, key, pk, ws, z, h):
        """The whole wg_inverse call (~250 launches) captured once per shape into a hipGraph (torch.cuda.CUDAGraph on the stream the C ABI
        enqueues on) and replayed; the packed weights and the workspace are referenced by address, so a re-pack in place is seen."""
        ent = self._graphs.get(key)
        if ent is None:
            sz, sh = z.clone(), h.clone()
            self._launch(pk, ws, sz, sh, True)                      # warm-up outside capture
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls are illegal during the capture -- another thread of the process (a serving thread,
            # an nn.DataParallel replica sharing this engine) that allocates or launches meanwhile is left alone; one capture at a time
            with _CAPTURE_LOCK, torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out, logdet = self._launch(pk, ws, sz, sh, True)
            while len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = (graph, sz, sh, out, logdet)
        graph, sz, sh, out, logdet = ent
        sz.copy_(z)
        sh.copy_(h)
        graph.replay()
        return out.clone(), logdet.clone()

    @on_device
    def backward(self, params, z, h, dz, dlogdet, need, need_dh, need_dx, want_x=False, grads_out=None, flow_events=None,
                 kept=None):
        """need[i]: produce the gradient of params[i] (into grads_out[i] when given).  Returns (grads, dh, dx, x_rebuilt).
        kept: what run_keep returned for this z; still-valid stored activations are used, overwritten ones fall back to
        the recomputing backward (same gradients, one extra WN forward per flow)."""
        require_device(z, h, dz, dlogdet)
        z, h, dz, dlogdet = z.contiguous(), h.contiguous(), dz.contiguous(), dlogdet.contiguous()
        B, N = z.shape
        F = h.shape[2]
        pk = self._pack(params, z.device)
        cfg = self.cfg
        if kept is not None and kept[0] == self._kept_gen:
            cfg, ws = self.cfg_keep, kept[1]
            self._kept_gen += 1                             # the backward rebuilds X in place: the stored state is spent
        else:
            ws = self._ws(B, N, 1, z.device)
        if grads_out is not None:
            grads = [g if nd else None for g, nd in zip(grads_out, need)]
        else:
            grads = [torch.empty_like(p) if (p is not None and nd) else None for p, nd in zip(params, need)]
        dh = torch.empty_like(h) if need_dh else None
        dx = torch.empty_like(z) if need_dx else None
        xr = torch.empty_like(z) if want_x else None
        check(_lib.lib().wg_backward(C.byref(cfg), _table(params), _p(pk), _p(z), _p(h), _p(dz), _p(dlogdet), B, N, F,
                                     _table(grads), _p(dh), _p(dx), _p(xr), _p(ws), ws.numel(), _stream(),
                                     self._events(flow_events)), "wg_backward")
        return grads, dh, dx, xr

    @on_device
    def train_step(self, params, x, h, sigma, elementwise_mean, need, grads_out=None, need_dh=False, flow_events=None,
                   keep=False, metrics=None):
        """wg_train_step: forward + NLL + backward in one call (the forward keeps the last flow's layers for the backward;
        keep=True -- memory_efficient=False -- keeps every flow's, so no WN is recomputed).
        metrics (optional, 4 device floats): receives the logged scalars [logdet/numel, z.mean, z.std, loss] (lightning.py:58-64).
        Returns (loss, z, logdet, grads, dh)."""
        require_device(x, h, *params)
        x, h = x.contiguous(), h.contiguous()
        B, N = x.shape
        pk = self._pack(params, x.device)
        cfg = self.cfg_keep if keep else self.cfg
        if keep:
            ws = self._ws_keep(B, N, x.device)
            self._kept_gen += 1
        else:
            ws = self._ws(B, N, 1, x.device)
        scratch = self.buffers.get((x.device, "step", B, N), 4 * _lib.lib().wg_train_scratch_floats(B, N), x.device)
        if grads_out is not None:
            grads = [g if nd else None for g, nd in zip(grads_out, need)]
        else:
            grads = [torch.empty_like(p) if (p is not None and nd) else None for p, nd in zip(params, need)]
        z = torch.empty_like(x)
        logdet = torch.empty(B, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dh = torch.empty_like(h) if need_dh else None
        check(_lib.lib().wg_train_step(C.byref(cfg), _table(params), _p(pk), _p(x), _p(h), B, N, h.shape[2], float(sigma),
                                       int(elementwise_mean), _p(z), _p(logdet), _p(loss), _p(metrics), _table(grads), _p(dh), _p(scratch),
                                       _p(ws), ws.numel(), _stream(), self._events(flow_events)), "wg_train_step")
        return loss, z, logdet, grads, dh

    @staticmethod
    def _events(events):
        """torch.cuda.Event list -> array of raw hipEvent_t (NULL when not given)"""
        if not events:
            return None
        arr = (C.c_void_p * len(events))()
        for i, e in enumerate(events):
            arr[i] = e.cuda_event
        return arr

    @on_device
    def upsample(self, params, h, T):
        require_device(h)
        h = h.contiguous()
        B, M, F = h.shape
        pk = self._pack(params, h.device)
        y = torch.empty(B, M, T, dtype=torch.float32, device=h.device)
        check(_lib.lib().wg_upsample(C.byref(self.cfg), _p(pk), _p(h), B, F, T, _p(y), _stream()), "wg_upsample")
        return y


class CouplingEngine:
    """Block-level entry points for AffineCouplingBlock(WN) / WN."""

    def __init__(self, dims: WgWnDims):
        self.dims = dims
        self.buffers = _Buffers()
        self.packed = PackedWeights()

    def _pack(self, params, device):
        L = _lib.lib()
        if self.packed.stale(params) or self.packed.buf.device != device:
            nbytes = L.wg_wn_packed_bytes(C.byref(self.dims))
            if nbytes == 0:
                raise WgError("WN configuration not supported by the HIP kernels "
                              "(channels must be multiples of 32, odd radix <= 9, in_channels <= 16)")
            if self.packed.buf is None or self.packed.buf.numel() < nbytes or self.packed.buf.device != device:
                self.packed.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            check(L.wg_wn_pack_weights(C.byref(self.dims), _table(params), _p(self.packed.buf), _stream()), "wg_wn_pack_weights")
            self.packed.commit()
        return self.packed.buf

    def _ws(self, B, T, mode, device):
        nbytes = _lib.lib().wg_coupling_workspace_bytes(C.byref(self.dims), B, T, mode)
        return self.buffers.get((device, mode, B, T), nbytes, device)

    @on_device
    def apply(self, params, x, y, reverse):
        require_device(x, y, *params)
        x, y = x.contiguous(), y.contiguous()
        B, c, T = x.shape
        if y.shape[2] != T or y.shape[1] != self.dims.aux_ch or c != 2 * self.dims.in_ch:
            raise WgError("coupling: x %s / y %s do not match the block's dimensions" % (tuple(x.shape), tuple(y.shape)))
        pk = self._pack(params, x.device)
        ws = self._ws(B, T, 0, x.device)
        z = torch.empty_like(x)
        log_s = torch.empty(B, c // 2, T, dtype=torch.float32, device=x.device)
        check(_lib.lib().wg_coupling_apply(C.byref(self.dims), _p(pk), _p(x), _p(y), B, T, int(reverse), _p(z), _p(log_s),
                                           _p(ws), ws.numel(), _stream()), "wg_coupling_apply")
        return z, log_s

    @on_device
    def wn(self, params, x, y):
        require_device(x, y, *params)
        x, y = x.contiguous(), y.contiguous()
        B, ic, T = x.shape
        pk = self._pack(params, x.device)
        ws = self._ws(B, T, 0, x.device)
        log_s, t = torch.empty_like(x), torch.empty_like(x)
        check(_lib.lib().wg_wn_apply(C.byref(self.dims), _p(pk), _p(x), _p(y), B, T, _p(log_s), _p(t), _p(ws), ws.numel(),
                                     _stream()), "wg_wn_apply")
        return log_s, t

    @on_device
    def backward(self, params, z, y, dz, dlog_s, reverse, need, need_dy, x_out):
        require_device(z, y, dz, dlog_s)
        z, y, dz, dlog_s = z.contiguous(), y.contiguous(), dz.contiguous(), dlog_s.contiguous()
        B, c, T = z.shape
        pk = self._pack(params, z.device)
        ws = self._ws(B, T, 1, z.device)
        grads = [torch.empty_like(p) if (p is not None and nd) else None for p, nd in zip(params, need)]
        dx = torch.empty_like(z)
        dy = torch.empty_like(y) if need_dy else None
        check(_lib.lib().wg_coupling_backward(C.byref(self.dims), _table(params), _p(pk), _p(z), _p(y), _p(dz), _p(dlog_s), B, T,
                                              int(reverse), _p(x_out), _p(dx), _p(dy), _table(grads), _p(ws), ws.numel(),
                                              _stream()), "wg_coupling_backward")
        return dx, dy, grads


_INV_BUFFERS = _Buffers()
_LAYER_BUFFERS = _Buffers()


@on_device
def layer_apply(dims, params, x, y):
    """NonCausalLayer.forward on its own (model/waveglow.py:41-46; wg_layer_apply): params = [W.g or None, W.v, W_o.g or None, W_o.v].
    Returns (res or None, skip)."""
    require_device(x, y, *params)
    x, y = x.contiguous(), y.contiguous()
    if dims.rows > 0:                                        # NonCausalLayer2D: x [B, C, H, W], y [B, 2 Cd, 1, W]
        B, Cin, H, T = x.shape
        if Cin != dims.res_ch or H != dims.rows or tuple(y.shape) != (B, 2 * dims.dil_ch, 1, T):
            raise WgError("NonCausalLayer2D: x %s / y %s do not match the layer (x [B, %d, H, W], y [B, %d, 1, W])"
                          % (tuple(x.shape), tuple(y.shape), dims.res_ch, 2 * dims.dil_ch))
    else:
        B, Cin, T = x.shape
        if Cin != dims.res_ch or tuple(y.shape) != (B, 2 * dims.dil_ch, T):
            raise WgError("NonCausalLayer: x %s / y %s do not match the layer (x [B, %d, T], y [B, %d, T])"
                          % (tuple(x.shape), tuple(y.shape), dims.res_ch, 2 * dims.dil_ch))
    nbytes = _lib.lib().wg_layer_workspace_bytes(C.byref(dims), B, T)
    if nbytes == 0:
        raise WgError("NonCausalLayer shape not supported by the HIP kernels (residual / skip channels multiples of 16, dilation channels "
                      "a multiple of 32, odd radix <= 9)")
    ws = _LAYER_BUFFERS.get((x.device, dims.res_ch, dims.dil_ch, dims.skip_ch, dims.radix, dims.dilation, dims.h_dilation, dims.rows, B, T), nbytes, x.device)
    res = None if dims.last_layer else torch.empty_like(x)
    skip = torch.empty((B, dims.skip_ch) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    check(_lib.lib().wg_layer_apply(C.byref(dims), _table([None if p is None else p.contiguous() for p in params]), _p(x), _p(y), B, T, _p(res),
                                    _p(skip), _p(ws), ws.numel(), _stream()), "wg_layer_apply")
    return res, skip


@on_device
def affine_apply(xb, log_s, t, reverse):
    """The coupling's own arithmetic on plain tensors (wg_affine_apply): xb * exp(log_s) + t, or (reverse) (xb - t) / exp(log_s)."""
    require_device(xb, log_s, t)
    xb, log_s, t = xb.contiguous(), log_s.contiguous(), t.contiguous()
    if not (xb.shape == log_s.shape == t.shape) or xb.dtype != torch.float32 or log_s.dtype != torch.float32 or t.dtype != torch.float32:
        raise WgError("affine coupling: the transform must return float32 (log_s, t) of the shape of the passed half, got %s / %s for %s"
                      % (tuple(log_s.shape), tuple(t.shape), tuple(xb.shape)))
    out = torch.empty_like(xb)
    check(_lib.lib().wg_affine_apply(_p(xb), _p(log_s), _p(t), xb.numel(), int(reverse), _p(out), _stream()), "wg_affine_apply")
    return out


@on_device
def affine_backward(out_half, log_s, t, dout, dlog_s, reverse):
    """wg_affine_backward -> (input half rebuilt, d/d log_s, d/d t of the transform's outputs, gradient of the input half)."""
    require_device(out_half, log_s, t, dout)
    out_half, log_s, t, dout = out_half.contiguous(), log_s.contiguous(), t.contiguous(), dout.contiguous()
    dlog_s = None if dlog_s is None else dlog_s.contiguous()
    rebuilt, g_ls, g_t, din = (torch.empty_like(out_half) for _ in range(4))
    check(_lib.lib().wg_affine_backward(_p(out_half), _p(log_s), _p(t), _p(dout), _p(dlog_s), out_half.numel(), int(reverse), _p(rebuilt),
                                        _p(g_ls), _p(g_t), _p(din), _stream()), "wg_affine_backward")
    return rebuilt, g_ls, g_t, din


@on_device
def layer_backward(dims, params, x, y, dres, dskip, need, need_dx, need_dy):
    """wg_layer_backward: (dx, dy, [dW.g, dW.v, dW_o.g, dW_o.v]) of NonCausalLayer / NonCausalLayer2D called on its own; None where not needed."""
    require_device(x, y, dskip, *params)
    x, y, dskip = x.contiguous(), y.contiguous(), dskip.contiguous()
    dres = None if dres is None else dres.contiguous()
    B, T = x.shape[0], x.shape[-1]
    nbytes = _lib.lib().wg_layer_backward_workspace_bytes(C.byref(dims), B, T)
    if nbytes == 0:
        raise WgError("NonCausalLayer shape not supported by the HIP kernels")
    ws = _LAYER_BUFFERS.get((x.device, "bwd", dims.res_ch, dims.dil_ch, dims.skip_ch, dims.radix, dims.dilation, dims.h_dilation, dims.rows, B, T), nbytes, x.device)
    params = [None if p is None else p.contiguous() for p in params]
    grads = [torch.empty_like(p) if (p is not None and nd) else None for p, nd in zip(params, need)]
    dx = torch.empty_like(x) if need_dx else None
    dy = torch.empty_like(y) if need_dy else None
    check(_lib.lib().wg_layer_backward(C.byref(dims), _table(params), _p(x), _p(y), _p(dres), _p(dskip), B, T, _p(dx), _p(dy), _table(grads),
                                       _p(ws), ws.numel(), _stream()), "wg_layer_backward")
    return dx, dy, grads


class LayerFn(torch.autograd.Function):
    """NonCausalLayer / NonCausalLayer2D forward on its own as an autograd node: wg_layer_apply / wg_layer_backward."""

    @staticmethod
    def forward(ctx, x, y, dims, *params):
        res, skip = layer_apply(dims, [None if p is None else p.detach() for p in params], x.detach(), y.detach())
        ctx.dims, ctx.n = dims, len(params)
        ctx.save_for_backward(x, y, *[p for p in params if p is not None])
        ctx.present = [p is not None for p in params]
        return (skip,) if res is None else (res, skip)

    @staticmethod
    def backward(ctx, *douts):
        x, y = ctx.saved_tensors[:2]
        it = iter(ctx.saved_tensors[2:])
        params = [next(it) if here else None for here in ctx.present]
        dres, dskip = (None, douts[0]) if len(douts) == 1 else douts
        need = [p is not None and ctx.needs_input_grad[3 + k] for k, p in enumerate(params)]
        dx, dy, grads = layer_backward(ctx.dims, [None if p is None else p.detach() for p in params], x.detach(), y.detach(), dres, dskip, need,
                                       ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (dx, dy, None) + tuple(grads)


@on_device
def invconv_apply(W, x, reverse):
    require_device(W, x)
    x = x.contiguous()
    B, c, T = x.shape
    Wm = W.reshape(c, c).contiguous()
    nbytes = _lib.lib().wg_invconv_workspace_bytes(c, B, T)
    ws = _INV_BUFFERS.get((x.device, c, B, T), nbytes, x.device)
    z = torch.empty_like(x)
    logdet = torch.empty((), dtype=torch.float32, device=x.device)
    check(_lib.lib().wg_invconv_apply(_p(Wm), c, _p(x), B, T, int(reverse), _p(z), _p(logdet), _p(ws), ws.numel(), _stream()),
          "wg_invconv_apply")
    return z, logdet


@on_device
def invconv_backward(W, z, dz, dlogdet, reverse, x_out):
    require_device(W, z, dz, dlogdet)
    z, dz = z.contiguous(), dz.contiguous()
    B, c, T = z.shape
    Wm = W.reshape(c, c).contiguous()
    nbytes = _lib.lib().wg_invconv_workspace_bytes(c, B, T)
    ws = _INV_BUFFERS.get((z.device, c, B, T), nbytes, z.device)
    dx = torch.empty_like(z)
    dW = torch.empty(c, c, dtype=torch.float32, device=z.device)
    check(_lib.lib().wg_invconv_backward(_p(Wm), c, _p(z), _p(dz), _p(dlogdet.contiguous()), B, T, int(reverse), _p(x_out), _p(dx),
                                         _p(dW), _p(ws), ws.numel(), _stream()), "wg_invconv_backward")
    return dx, dW


@on_device
def nll_loss(z, logdet, sigma, elementwise_mean, metrics=None):
    """loss (device scalar); metrics (optional, 4 device floats) receives [logdet.sum()/z.numel(), z.mean(), z.std(), loss]."""
    require_device(z, logdet, metrics)
    z, logdet = z.contiguous(), logdet.contiguous()
    B, N = z.shape
    loss = torch.empty((), dtype=torch.float32, device=z.device)
    scratch = torch.empty(_lib.lib().wg_nll_scratch_floats(B), dtype=torch.float32, device=z.device)
    check(_lib.lib().wg_nll_loss(_p(z), _p(logdet), B, N, float(sigma), int(elementwise_mean), _p(loss), _p(metrics), _p(scratch),
                                 _stream()), "wg_nll_loss")
    return loss


def training_metrics(z, logdet, sigma=1.0, elementwise_mean=True):
    """The scalars LightModel.training_step logs (model/lightning.py:58-64) as ONE device vector
    [logdet.sum() / z.numel(), z.mean(), z.std(), loss]; a data-parallel caller mean-reduces it (sync_dist=True)."""
    m = torch.empty(4, dtype=torch.float32, device=z.device)
    nll_loss(z, logdet, sigma, elementwise_mean, metrics=m)
    return m


@on_device
def nll_loss_backward(z, sigma, elementwise_mean, dloss):
    require_device(z, dloss)
    z = z.contiguous()
    B, N = z.shape
    dz = torch.empty_like(z)
    dlogdet = torch.empty(B, dtype=torch.float32, device=z.device)
    check(_lib.lib().wg_nll_loss_backward(_p(z), B, N, float(sigma), int(elementwise_mean), _p(dloss.contiguous()), _p(dz),
                                          _p(dlogdet), _stream()), "wg_nll_loss_backward")
    return dz, dlogdet


# ---- WSRGlow conditioning front-end (include/wgflow.h: wg_wsr_cond*) -----------------------------------------------------
WSR_COND_CHANNELS = 8 * 400 + 9 * 51


@on_device
def wsr_cond(c, mu_table, ang_table):
    """c[B,L] -> cond[B,3659,L/8]  (WSRGlow._get_cond, model/wsrglow.py:37-50; c is read clipped, not modified)."""
    require_device(c, mu_table, ang_table)
    if c.dim() != 2 or c.size(1) % 8 or c.size(1) < 8:
        raise WgError("WSRGlow conditioning signal must be [B, L] with L a multiple of 8")
    if tuple(mu_table.shape) != (256, 400) or tuple(ang_table.shape) != (120, 50):
        raise WgError("WSRGlow embedding tables must be [256,400] and [120,50]")
    c, mu_table, ang_table = c.contiguous(), mu_table.contiguous(), ang_table.contiguous()
    B, L = c.shape
    cond = torch.empty(B, WSR_COND_CHANNELS, L // 8, dtype=torch.float32, device=c.device)
    check(_lib.lib().wg_wsr_cond(_p(c), B, L, _p(mu_table), _p(ang_table), _p(cond), _stream()), "wg_wsr_cond")
    return cond


@on_device
def wsr_cond_pre(c):
    """Diagnostics (wg_wsr_cond_pre): the quantisers' float32 values before truncation -> (mu_pre[B,L], ang_pre[B,9,L/8])."""
    require_device(c)
    c = c.contiguous()
    B, L = c.shape
    mu_pre = torch.empty(B, L, dtype=torch.float32, device=c.device)
    ang_pre = torch.empty(B, 9, L // 8, dtype=torch.float32, device=c.device)
    check(_lib.lib().wg_wsr_cond_pre(_p(c), B, L, _p(mu_pre), _p(ang_pre), _stream()), "wg_wsr_cond_pre")
    return mu_pre, ang_pre


@on_device
def wsr_cond_backward(c, dcond, out=None):
    """-> (d mu_table [256,400], d ang_table [120,50]) from dcond[B,3659,L/8]; out: optional pair of contiguous tensors to fill."""
    require_device(c, dcond)
    c, dcond = c.contiguous(), dcond.contiguous()
    B, L = c.shape
    if tuple(dcond.shape) != (B, WSR_COND_CHANNELS, L // 8):
        raise WgError("dcond must be [B, 3659, L/8]")
    if out is not None:
        dmu, dang = out
        require_device(dmu, dang)
        if tuple(dmu.shape) != (256, 400) or tuple(dang.shape) != (120, 50) or not (dmu.is_contiguous() and dang.is_contiguous()):
            raise WgError("wsr_cond_backward: out must be contiguous [256,400] and [120,50] tensors")
    else:
        dmu = torch.empty(256, 400, dtype=torch.float32, device=c.device)
        dang = torch.empty(120, 50, dtype=torch.float32, device=c.device)
    check(_lib.lib().wg_wsr_cond_backward(_p(c), B, L, _p(dcond), _p(dmu), _p(dang), _stream()), "wg_wsr_cond_backward")
    return dmu, dang


# ---- WaveFlow (include/wgflow.h: wg_wf_*) ------------------------------------------------------------------------------------
class WaveFlowEngine:
    """Whole-model entry points of WaveFlow: wg_wf_pack_weights / wg_wf_forward / wg_wf_inverse / wg_wf_backward."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.buffers = _Buffers()
        self.packed = PackedWeights()
        self._tape = None

    def _pack(self, params, device):
        L = _lib.lib()
        if len(params) != L.wg_wf_param_count(C.byref(self.cfg)):
            raise WgError("WaveFlow parameter table has %d entries" % len(params))
        if self.packed.stale(params) or self.packed.buf.device != device:
            require_device(*params)
            nbytes = L.wg_wf_packed_bytes(C.byref(self.cfg))
            if nbytes == 0:
                raise WgError("WaveFlow configuration not supported by the HIP kernels (n_group in {8,16,32,64,128}, channels "
                              "multiples of 32)")
            if self.packed.buf is None or self.packed.buf.numel() < nbytes or self.packed.buf.device != device:
                self.packed.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            check(L.wg_wf_pack_weights(C.byref(self.cfg), _table(params), _p(self.packed.buf), _stream()), "wg_wf_pack_weights")
            self.packed.commit()
        return self.packed.buf

    def _ws(self, B, N, mode, device):
        nbytes = _lib.lib().wg_wf_workspace_bytes(C.byref(self.cfg), B, N, mode)
        return self.buffers.get((device, mode, B, N), nbytes, device)

    @on_device
    def forward(self, params, x, mel, keep_tape):
        require_device(x, mel, *params)
        x, mel = x.contiguous(), mel.contiguous()
        B, N = x.shape
        pk = self._pack(params, x.device)
        ws = self._ws(B, N, 0, x.device)
        tape = None
        if keep_tape:
            # one tape PER forward call, owned by that call's autograd node: a second forward of the same shape before the first
            # backward (two losses, gradient accumulation, a grad-enabled evaluation pass) must not overwrite the flow inputs the
            # first backward recomputes from.  (zero-filled: the kernels keep the halo of every plane zero, they do not write it)
            nbytes = _lib.lib().wg_wf_tape_bytes(C.byref(self.cfg), B, N)
            tape = torch.zeros(nbytes, dtype=torch.uint8, device=x.device)
        z = torch.empty_like(x)
        logdet = torch.empty(B, dtype=torch.float32, device=x.device)
        check(_lib.lib().wg_wf_forward(C.byref(self.cfg), _table(params), _p(pk), _p(x), _p(mel), B, N, mel.shape[2], _p(z), _p(logdet),
                                       _p(tape), _p(ws), ws.numel(), _stream()), "wg_wf_forward")
        return z, logdet, tape

    @on_device
    def upsample(self, params, mel):
        """WaveFlow._upsample_h: mel [B, n_mels, F] -> [B, n_mels, F s - 2 (s // 2) + 2 s + 1]"""
        require_device(mel, *params)
        mel = mel.contiguous()
        B, M, F = mel.shape
        s = 256 // self.cfg.n_group
        T = F * s - 2 * (s // 2) + 2 * s + 1
        pk = self._pack(params, mel.device)
        y = torch.empty(B, M, T, dtype=torch.float32, device=mel.device)
        check(_lib.lib().wg_wf_upsample(C.byref(self.cfg), _table(params), _p(pk), _p(mel), B, F, T, _p(y), _stream()), "wg_wf_upsample")
        return y

    @on_device
    def inverse(self, params, z, mel):
        require_device(z, mel, *params)
        z, mel = z.contiguous(), mel.contiguous()
        B, N = z.shape
        pk = self._pack(params, z.device)
        ws = self._ws(B, N, 1, z.device)
        x = torch.empty_like(z)
        logdet = torch.empty(B, dtype=torch.float32, device=z.device)
        check(_lib.lib().wg_wf_inverse(C.byref(self.cfg), _table(params), _p(pk), _p(z), _p(mel), B, N, mel.shape[2], _p(x), _p(logdet),
                                       _p(ws), ws.numel(), _stream()), "wg_wf_inverse")
        return x, logdet

    @on_device
    def backward(self, params, tape, mel, dz, dlogdet, need_dmel, need_dx):
        require_device(mel, dz, dlogdet)
        mel, dz, dlogdet = mel.contiguous(), dz.contiguous(), dlogdet.contiguous()
        B, N = dz.shape
        pk = self._pack(params, dz.device)
        ws = self._ws(B, N, 1, dz.device)
        grads = [None if p is None else torch.empty_like(p) for p in params]
        dmel = torch.empty_like(mel) if need_dmel else None
        dx = torch.empty_like(dz) if need_dx else None
        check(_lib.lib().wg_wf_backward(C.byref(self.cfg), _table(params), _p(pk), _p(tape), _p(mel), _p(dz), _p(dlogdet), B, N,
                                        mel.shape[2], _table(grads), _p(dmel), _p(dx), _p(ws), ws.numel(), _stream()), "wg_wf_backward")
        return grads, dmel, dx


class WN2DEngine(WaveFlowEngine):
    """WN2D.forward on its own (wg_wf_wn_apply): the packing and workspaces of a one-flow WaveFlow whose upsampler entries are placeholders."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self._dummy = None

    def table(self, wn_params, device):
        if self._dummy is None or self._dummy[0].device != device:
            M, s = self.cfg.n_mels, 256 // self.cfg.n_group
            self._dummy = [torch.zeros(M, device=device), torch.ones(M, M, 2 * s + 1, device=device)]
        return [self._dummy[0], None, self._dummy[1]] + list(wn_params)

    @on_device
    def apply(self, wn_params, x, y):
        require_device(x, y, *wn_params)
        B, one, rows, W = x.shape
        if one != 1 or rows > self.cfg.n_group or tuple(y.shape) != (B, self.cfg.n_mels, W):
            raise WgError("WN2D: x %s / y %s do not match (x [B, 1, rows <= %d, W], y [B, %d, W])"
                          % (tuple(x.shape), tuple(y.shape), self.cfg.n_group, self.cfg.n_mels))
        x, y = x.contiguous(), y.contiguous()
        params = self.table(wn_params, x.device)
        pk = self._pack(params, x.device)
        ws = self._ws(B, W * self.cfg.n_group, 0, x.device)
        log_s, t = torch.empty_like(x), torch.empty_like(x)
        check(_lib.lib().wg_wf_wn_apply(C.byref(self.cfg), _table(params), _p(pk), _p(x), _p(y), B, rows, W, _p(log_s), _p(t), _p(ws),
                                        ws.numel(), _stream()), "wg_wf_wn_apply")
        return log_s, t

    @on_device
    def backward(self, wn_params, x, y, dlog_s, dt, need, need_dx, need_dy):
        """wg_wf_wn_backward: gradients of x / y (None where not needed) and of the WN2D parameters (`need`: per table entry)."""
        require_device(x, y, dlog_s, dt, *wn_params)
        x, y, dlog_s, dt = x.contiguous(), y.contiguous(), dlog_s.contiguous(), dt.contiguous()
        B, one, rows, W = x.shape
        params = self.table(wn_params, x.device)
        pk = self._pack(params, x.device)
        ws = self._ws(B, W * self.cfg.n_group, 1, x.device)
        grads = [None, None, None] + [torch.empty_like(p) if (p is not None and nd) else None for p, nd in zip(wn_params, need)]
        dx = torch.empty_like(x) if need_dx else None
        dy = torch.empty_like(y) if need_dy else None
        check(_lib.lib().wg_wf_wn_backward(C.byref(self.cfg), _table(params), _p(pk), _p(x), _p(y), _p(dlog_s), _p(dt), B, rows, W, _p(dx), _p(dy),
                                           _table(grads), _p(ws), ws.numel(), _stream()), "wg_wf_wn_backward")
        return dx, dy, grads[3:]


# ---- log-mel conditioner (include/wgflow.h: wg_melspec) -------------------------------------------------------------------------
@on_device
def melspec(x, sr, n_fft, hop, f_min, f_max, n_mels, return_power=False):
    """audio [B, N] -> log-mel [B, n_mels, N // hop + 1]  (MelSpec.forward, model/condition.py:18-19).
    return_power: also the power spectrogram [B, n_fft // 2 + 1, frames] the mel filters are applied to."""
    require_device(x)
    if x.dim() != 2:
        raise WgError("MelSpec expects audio [B, N]")
    x = x.contiguous()
    B, N = x.shape
    frames = _lib.lib().wg_melspec_frames(N, n_fft, hop)
    out = torch.empty(B, n_mels, frames, dtype=torch.float32, device=x.device)
    power = torch.empty(B, n_fft // 2 + 1, frames, dtype=torch.float32, device=x.device) if return_power else None
    check(_lib.lib().wg_melspec(_p(x), B, N, int(sr), int(n_fft), int(hop), float(f_min), float(f_max if f_max is not None else 0.0),
                                int(n_mels), _p(out), _p(power), _stream()), "wg_melspec")
    return (out, power) if return_power else out


@on_device
def lowpass(x, n_fft, hop, cut_bins, step):
    """x [B, T] -> STFT low-pass (bins >= cut_bins zeroed), every step-th sample: [B, ceil(T / step)]  (condition.py:22-66)."""
    require_device(x)
    if x.dim() != 2:
        raise WgError("LowPass expects audio [B, T]")
    x = x.contiguous()
    B, T = x.shape
    nbytes = _lib.lib().wg_lowpass_workspace_bytes(B, T, int(n_fft), int(hop))
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=x.device)
    out = torch.empty(B, (T + step - 1) // step, dtype=torch.float32, device=x.device)
    check(_lib.lib().wg_lowpass(_p(x), B, T, int(n_fft), int(hop), int(cut_bins), int(step), _p(out), _p(ws), ws.numel(), _stream()), "wg_lowpass")
    return out
