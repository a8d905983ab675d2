#!/usr/bin/env python3
"""Headline benchmark: audio samples/s of WaveGlow-256ch forward + NLL backward on 16 000-sample segments.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu] [--no-inverse] [--spawn]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is only a LAUNCHER.  Before anything touches the GPU it
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (one RCCL rank per GPU, per-GPU batch fixed: the
reference's `batch_size //= gpus` read the other way round, train.py:51-53,73-78), relays the ranks' output to stderr and prints rank 0's
JSON line as the last line of stdout; a failing child is a failing bench.  Under torch.distributed.run (WORLD_SIZE set) it is a rank, and
WORLD_SIZE must equal --gpus.  `--spawn` forces the launcher for N = 1 as well (a 1-rank RCCL group through the same code path).

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
z, logdet = model(x, h); loss = NLL(z, logdet); backward to all 459 parameter gradients (+ RCCL mean all-reduce
of the gradients when N > 1).  Data loading, mel computation, optimizer step and logging are outside, as in
SURVEY.md 8d.  Per-GPU batch is fixed (weak scaling): configs[1] of BASELINE.json at N=1, configs[2] at N=8.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant kernel (dilated conv + conditioning + gate; the line names the instantiation that RAN, as the library
                 reports it per launch: wg_timer_read_name) timed with HIP events over as many EXTRA steps behind the timed ones
                 (nothing is attached while the headline runs), algorithmic FLOPs / average launch duration vs the dense bf16 MFMA peak
                 (2.5 PF; only algorithmic FLOPs are credited, the 3 issued bf16 products per fp32 product are overhead;
                 `f32_mode` carries the exact-fp32 run against the 157.3 TF fp32 MFMA peak);
  cpu_baseline : the CPU path (oracle/torch_cpu.py: the algorithm restated on ATen's CPU kernels, the library the reference runs
                 on) timed on the host cores on one 16 000-sample segment of the same workload (rank 0, N=1 only), with the plain-C
                 oracle next to it as `c_port`;
  inverse_khz  : single-GPU synthesis speed, timed as the reference does (inference.py:50-56);
  box          : a fixed matrix-pipe + LDS loop of the library (wg_box_probe, ~0.5 s before the warm-up is over): issued TFLOP/s and the
                 in-kernel clock of THIS GPU -- the boxes of the pool differ by a few per cent, this makes that a measured field.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2 = dict(flows=12, n_group=8, n_early_every=4, n_early_size=2, hop_size=256, n_mels=80,
          dilation_channels=256, residual_channels=256, skip_channels=256, depth=8, radix=3)   # configs/waveglow_LJ_speech.json:6-19
SEG, FRAMES, SIGMA = 16000, 63, 0.7
# the config behind the ONLY speed the reference publishes ("around 470 kHz on a 1080ti", README.md:64-67): configs/musicnet_config.json:7-20
MUSICNET = dict(flows=18, n_group=8, n_early_every=6, n_early_size=2, hop_size=512, n_mels=80,
                dilation_channels=256, residual_channels=256, skip_channels=256, depth=4, radix=3)
MUSICNET_PUBLISHED_KHZ = 470.0            # /root/reference/README.md:67 (GTX 1080 Ti)
FP32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0           # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)
FWD_FLOP_PER_SAMPLE = 13_376_372         # SURVEY.md 2.1 (forward = inverse)
STEP_FLOP_PER_SAMPLE = 40_129_116        # fwd + dgrad + wgrad (algorithmic; the recompute is not credited)


def fwd_flop_per_sample(cfg):
    """Multiply-adds x 2 of one forward (= inverse) pass per audio sample, from the architecture (SURVEY.md 2.1 / 8d's count: start,
    conditioning, the dilated convs, W_o, end and the 1x1 conv of every flow; c = channels left after the early outputs)."""
    C, Cd, Cs, D, R = cfg["residual_channels"], cfg["dilation_channels"], cfg["skip_channels"], cfg["depth"], cfg["radix"]
    total, c = 0, cfg["n_group"]
    for k in range(cfg["flows"]):
        if k and k % cfg["n_early_every"] == 0:
            c -= cfg["n_early_size"]
        total += 2 * (c // 2) * C + D * 2 * cfg["n_mels"] * 2 * Cd + D * 2 * R * C * 2 * Cd + (D - 1) * 2 * Cd * (C + Cs) + 2 * Cd * Cs + 2 * Cs * c + 2 * c * c
    return total / cfg["n_group"]


def build_model(dev, seed=0, cfg=None):
    import constant_memory_waveglow_amd as cm
    torch.manual_seed(seed)
    model = cm.WaveGlow(memory_efficient=True, bias=False, **(cfg or C2))
    with torch.no_grad():                 # the reference zero-inits WN.end (log_s = t = 0); make the flow non-trivial
        for blk in model.WNs:
            blk.F.end.weight.normal_(0.0, 0.02)
    return model.to(dev)


EPI_NUM = {"EPI_STORE": 0, "EPI_GATE": 1, "EPI_RESSKIP": 2, "EPI_DGATE": 3, "EPI_STORE_SO": 4, "EPI_GATE_SO": 5, "EPI_DGATE_SO": 6,
           "EPI_STORE_FO": 7, "WGG_EPI_PART": 8}               # csrc/wg_gemm.h, csrc/wg_gemm16g.h


def launch_site_to_rocprof(expr):
    """The kernel expression of a launch site as the library records it (wg_timer_read_name: "(convgemm16q_kernel<EPI_GATE_SO, 2, 2>)")
    in the form rocprofv3 prints, WITHOUT the closing bracket and the defaulted template arguments the launch site leaves out
    ("convgemm16q_kernel<5, 2, 2"): a prefix of the rocprofv3 name ("void convgemm16q_kernel<5, 2, 2, false, false>(ConvGemm16sArgs)")."""
    e = expr.strip()
    while e.startswith("(") and e.endswith(")"):
        e = e[1:-1].strip()
    if "<" not in e:
        return e
    name, targs = e.split("<", 1)
    targs = targs.rsplit(">", 1)[0]
    parts = [str(EPI_NUM.get(a.strip(), a.strip())) for a in targs.split(",")]
    return "%s<%s" % (name.strip(), ", ".join(parts))


# template arguments a launch site may leave out, as the kernels declare them (csrc/wg_gemm16q.h convgemm16q_kernel<EPI, NI, MG = 1,
# M64 = false, CG2 = false>, csrc/wg_small.h end_affine_kernel<NR, SEAM = false>): (number of parameters, the trailing defaults)
TEMPLATE_DEFAULTS = {"convgemm16q_kernel": (5, ["1", "false", "false"]), "end_affine_kernel": (2, ["false"])}


def full_instantiation(site_prefix):
    """launch_site_to_rocprof's prefix padded with the kernel's declared defaults and closed: "convgemm16q_kernel<5, 2, 2" ->
    "convgemm16q_kernel<5, 2, 2, false, false>" -- the name rocprofv3 prints up to its argument list, to be matched EXACTLY (a bare
    prefix also matched "convgemm16q_kernel<5, 2, 2, false, true>", another instantiation launched from another site)."""
    if "<" not in site_prefix:
        return site_prefix
    name, targs = site_prefix.split("<", 1)
    parts = [a.strip() for a in targs.split(",")]
    n, tail = TEMPLATE_DEFAULTS.get(name, (len(parts), []))
    if len(parts) < n:
        parts += tail[len(tail) - (n - len(parts)):]
    return "%s<%s>" % (name, ", ".join(parts))


def _traffic(kernel_prefix, prefix=""):
    """(HBM bytes per launch of the dominant kernel, the file it was read from, the kernel's full name there).  NOT measured in this
    run: the bytes come from the newest committed PMC summary (profiles/*_hbm_traffic.json, produced by tools/profile_summary.py from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)
    whose kernel table holds EXACTLY ONE name that is the instantiation `kernel_prefix` stands for (launch_site_to_rocprof of the
    instantiation that ran in THIS run, defaulted template arguments filled in from the kernel's declaration: a summary of another
    kernel or of another instantiation is never cited); (None, None, None) if there is none, or if a summary is ambiguous."""
    import glob
    want = full_instantiation(kernel_prefix)
    best = (None, None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
        tag = os.path.basename(f)
        if (prefix and prefix not in tag) or (not prefix and ("_wf_" in tag or "_wsr_" in tag)):
            continue
        try:
            k = json.load(open(f))["kernels"]
        except Exception:
            continue
        hits = []
        for name, row in k.items():
            bare = name[5:] if name.startswith("void ") else name
            if bare == want or (bare.startswith(want) and bare[len(want):len(want) + 1] == "("):
                hits.append((row["hbm_bytes_per_launch"], os.path.relpath(f, ROOT), bare))
        if len(hits) == 1:
            best = hits[0]
        elif len(hits) > 1:
            best = (None, None, None)                          # (the newest summary is ambiguous: cite nothing rather than an older one)
    return best


def site_traffic(site, prefix=""):
    """_traffic for a launch site as the library names it.  A class entry that covers two launches is named "A + B" (the gate conv cut
    along K: partial products + the kernel that sums them and applies the gate): the bytes of both, and both names; (None, None, None)
    unless every part is in ONE summary."""
    parts = [launch_site_to_rocprof(p) for p in site.split(" + ") if p.strip()]
    if not parts:
        return None, None, None
    got = [_traffic(p, prefix) for p in parts]
    if any(g[0] is None for g in got) or len({g[1] for g in got}) != 1:
        return None, None, None
    return sum(g[0] for g in got), got[0][1], " + ".join(g[2] for g in got)


def box_probe(dev, ms=500):
    """`box`: the library's fixed matrix-pipe + LDS loop (wg_box_probe, csrc/wg_probe.h) on this GPU, ~0.5 s: issued TFLOP/s and the
    clock the chip holds in it.  The boxes of the pool differ by a few per cent on exactly this; with it in the line a reader can tell a
    slower box from a slower build."""
    from constant_memory_waveglow_amd import _lib
    L = _lib.lib()
    scratch = torch.empty(int(L.wg_box_probe_bytes()), dtype=torch.uint8, device=dev)
    out = (C.c_double * 3)()
    torch.cuda.synchronize()
    rc = L.wg_box_probe(scratch.data_ptr(), int(ms), out, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        return {"error": rc}
    return {"probe": "wg_box_probe: 256 workgroups x 8 waves, 64 x 64 tile per wave, hi / lo fragments re-read from LDS, 3 v_mfma_f32_16x16x32_bf16 "
                     "per fragment pair, random data, no global traffic; ~%d ms of back-to-back launches" % ms,
            "tflops_issued": out[0], "clock_ghz": out[1], "ms_per_launch": out[2]}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:                                         # noqa: BLE001
        pass
    return os.cpu_count() or 1


def cpu_baseline():
    """The CPU path timed on the host cores of the GPU box (rank 0, N = 1 only; BASELINE.md section 4's plan): training steps of
    oracle/torch_cpu.py -- the path's algorithm restated on the library the reference itself runs on (ATen's CPU convolutions, a local
    autograd graph per WN, activations rebuilt flow by flow), pinned to the reference's golden vectors by tests/test_oracle_golden.py --
    `value` = the C2 network on ALL cores the host grants (min(physical cores, cgroup CPU quota)): worker processes x threads in the split a
    short C1 probe finds fastest, one 16000-sample segment per process (batch items are independent units), 1 warm-up + 3 timed steps.  Next to it one process alone (`single_process`, C2 B=1,
    its fastest thread count) and C1 (64ch, 6 flows, B=2, seg 4000).  The plain-C oracle (oracle/wg_oracle.c, OpenMP), which the parity tests use as their
    checker, is timed next to it on C2 B=1 as `c_port`."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import fill
    from oracle import torch_cpu
    from oracle import wg_oracle as orc
    def timed(step, B, N, runs=3):
        step()                                                         # warm-up (page-in, thread pools, MKLDNN primitive cache)
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return {"samples_per_s": B * N / ts[len(ts) // 2], "median_s": ts[len(ts) // 2], "min_s": ts[0], "runs": runs, "batch": B}

    def case(cfg_name, cfg, B, N, F):
        specs = fill.model_param_specs(cfg)
        tab = fill.table(specs, fill.fill_params(specs, cfg_name + "/"))
        audio, h = fill.inputs(cfg_name + "/cpu%d" % B, B, N, F, cfg["n_mels"])
        return specs, tab, audio, h

    res = {}
    c1 = fill.CONFIGS["c1"]
    _, tab, audio, h = case("c1", c1, 2, 4000, 16)
    # ATen's intra-op pool does not scale with the core count on these shapes (T = 500 .. 2000 columns per convolution): on the 64-core
    # host of the GPU box all 64 threads ran the C2 step FIVE times slower than 8 threads do.  So the thread count is probed on the small
    # configuration (a fraction of a second per try) and the fastest is used; `cores` reports the threads actually used.
    phys = _physical_cores()
    probe = {}
    for n in sorted({min(phys, c) for c in (4, 8, 16, 32, phys)}):
        torch_cpu.set_threads(n)
        probe[n] = timed(lambda: torch_cpu.train_step(c1, tab, audio, h, SIGMA), 2, 4000, runs=2)["median_s"]
    cores = torch_cpu.set_threads(min(probe, key=probe.get))
    res["c1_b2"] = timed(lambda: torch_cpu.train_step(c1, tab, audio, h, SIGMA), 2, 4000)
    _, tab, audio, h = case("c2", C2, 1, SEG, FRAMES)
    res["c2_b1"] = timed(lambda: torch_cpu.train_step(C2, tab, audio, h, SIGMA), 1, SEG)
    # ALL cores: batch items are independent units, so the honest all-core figure is one segment per worker PROCESS, physical_cores // 8
    # processes x 8 threads (where ATen's pool still scales), all released together after a warm-up step, 3 timed steps each
    # (oracle/torch_cpu.py: time_parallel).  (A B = 2 step in ONE process ran 2.3x slower per sample than B = 1 in round 3 -- the thread
    # pool, not the algorithm -- and is no longer quoted.)
    # What "all cores" means on this host: the container may show 128 physical cores and grant a cgroup quota of 16 CPUs (the GPU boxes of
    # this pool do), and ATen's pool does not scale past a few threads on these shapes.  So the budget is min(physical cores, quota), and
    # the split of it into processes x threads is probed on C1 (2 timed steps per candidate) and the fastest runs the C2 measurement.
    logical, quota, _ = torch_cpu.host_cpu_budget()
    budget = max(1, min(phys, int(quota + 0.5)) if quota else phys)
    cands = sorted({(max(1, budget // t), t) for t in (8, 4, 2, 1) if budget // t >= 1} | {(min(2 * budget, phys), 1)})
    split_probe = {}
    c1_tab = fill.table(fill.model_param_specs(c1), fill.fill_params(fill.model_param_specs(c1), "c1/"))
    c1_audio, c1_h = fill.inputs("c1/cpu2", 2, 4000, 16, c1["n_mels"])
    for w_, t_ in cands:
        split_probe["%dx%d" % (w_, t_)] = torch_cpu.time_parallel(c1, c1_tab, c1_audio[:1], c1_h[:1], SIGMA, workers=w_, threads=t_, runs=2)["samples_per_s"]
    best = max(split_probe, key=split_probe.get)
    workers, wthreads = (int(v) for v in best.split("x"))
    allcore = torch_cpu.time_parallel(C2, tab, audio, h, SIGMA, workers=workers, threads=wthreads, runs=3)
    allcore.update({"cgroup_quota_cpus": quota, "logical_cpus": logical, "core_budget": budget, "split_probe_c1_samples_per_s": split_probe})
    total = sum(r["median_s"] * (r["runs"] + 1) for r in res.values()) + allcore["wall_s"] * 4.0 / 3.0
    # the C port: capped at 64 OpenMP threads (its loops expose 64-128 independent row blocks; more threads measured slower)
    oc = orc.make_config(**C2)
    cthreads = orc.set_threads(min(int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)), 64))
    cport = timed(lambda: orc.train_step(oc, tab, audio, h, SIGMA), 1, SEG, runs=2)
    cport.update({"unit": "samples/s", "cores": cthreads, "what": "oracle/wg_oracle.c (plain C + OpenMP), C2 B=1: 1 warm-up + 2 runs, median"})
    single = dict(res["c2_b1"], cores=cores, what="ONE process, its fastest thread count (probed on C1): C2 B=1, median of 3 runs after 1 warm-up")
    return {"value": allcore["samples_per_s"], "unit": "samples/s", "cores": min(budget, allcore["workers"] * allcore["threads_per_worker"]), "kind": "port",
            "implementation": "torch-cpu: oracle/torch_cpu.py, the step restated on ATen's CPU kernels (F.conv1d / MKLDNN), "
                              "constant-memory backward with a local autograd graph per WN; torch %s" % torch.__version__,
            "cpu_model": _cpu_model(),
            "sample": "WaveGlow-256ch 12 flows fwd+NLL+bwd, one 16000-sample segment per worker process: %d processes x %d threads, 1 warm-up + 3 "
                      "timed steps each, released together; value = processes x 3 x 16000 / (last end - first start).  Beside it: one "
                      "process alone (`single_process`), C1 (64ch, 6 flows, B=2, seg 4000) and the plain-C oracle; about %.0f s of CPU work in all"
                      % (allcore["workers"], allcore["threads_per_worker"], total + cport["median_s"] * 3),
            "all_core": allcore, "single_process": single, "configs": res, "c_port": cport,
            "thread_probe_c1_s": {str(k): v for k, v in probe.items()}, "physical_cores": phys,
            "build_container_reference": {
                "note": "measured in the BUILD container (8 cores of a Xeon, not this host), where the reference can be imported: the "
                        "reference's own torch-CPU path 7 980 samples/s (BASELINE.md section 2), oracle/torch_cpu.py 9 590, the C port "
                        "2 300, all on the C2 B=1 step -- context for how close the figures are to the reference's speed, not an extrapolation",
                "reference_samples_per_s": 7980.0, "torch_cpu_samples_per_s": 9590.0, "c_port_samples_per_s": 2300.0}}


def other_models(dev):
    """Secondary figures for the SURVEY.md 8f rows that are built (not the headline metric): one training step of WSRGlow 2x
    (configs/wsrglow_vctk_2x.json: batch 12 x 8192) and of WaveFlow (configs/waveflow_LJ_speech.json: batch 12 x 16000), forward +
    NLL + backward through autograd, 1 warm-up + 3 timed steps each.  A failure here never touches the headline line."""
    import constant_memory_waveglow_amd as cm
    res = {}

    last = {}

    def timed(step):
        last["loss"] = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            last["loss"] = step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3

    def checked(loss):
        """the secondary lines are only worth printing for a step that produced finite numbers"""
        v = float(loss)
        if not np.isfinite(v):
            raise FloatingPointError("non-finite loss %r" % v)
        return v

    try:
        torch.manual_seed(0)
        m = cm.WaveFlow(flows=8, n_group=64, n_mels=80, use_conv1x1=False, memory_efficient=False, dilation_channels=64,
                        residual_channels=64, skip_channels=64, bias=False)
        with torch.no_grad():
            for wn in m.WNs:
                wn.end.weight.normal_(0.0, 0.02)
        m = m.to(dev)
        crit = cm.WaveGlowLoss(1.0)
        x = torch.rand(12, 16000, device=dev) * 2 - 1
        h = torch.randn(12, 80, 63, device=dev)

        def step():
            m.zero_grad(set_to_none=True)
            m._engine.packed.key = None            # as in training, where the weights change: re-pack every step
            z, ld = m(x, h)
            loss = crit(z, ld)
            loss.backward()
            return loss
        dt = timed(step)
        gfin = all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())
        res["waveflow"] = {"workload": "WaveFlow 64ch 8 flows n_group 64, batch 12 x 16000, fwd+NLL+bwd", "ms_per_step": dt * 1e3,
                           "samples_per_s": 12 * 16000 / dt, "loss": checked(last["loss"]), "grads_finite": gfin}
        # the same step with the contractions on the exact-fp32 MFMA (the 2-D taps exist in all three arithmetic modes since round 3)
        old_prec = os.environ.get("WG_PRECISION")
        os.environ["WG_PRECISION"] = "f32"
        try:
            torch.manual_seed(0)
            m32 = cm.WaveFlow(flows=8, n_group=64, n_mels=80, use_conv1x1=False, memory_efficient=False, dilation_channels=64,
                              residual_channels=64, skip_channels=64, bias=False)
            m32.load_state_dict(m.state_dict())
            m32 = m32.to(dev)

            def step32():
                m32.zero_grad(set_to_none=True)
                m32._engine.packed.key = None
                z, ld = m32(x, h)
                loss = crit(z, ld)
                loss.backward()
                return loss
            dt32 = timed(step32)
            res["waveflow"]["f32_mode"] = {"ms_per_step": dt32 * 1e3, "samples_per_s": 12 * 16000 / dt32, "loss": checked(last["loss"])}
            del m32
        finally:
            if old_prec is None:
                os.environ.pop("WG_PRECISION", None)
            else:
                os.environ["WG_PRECISION"] = old_prec
        with torch.no_grad():                      # synthesis (row-by-row inverse): ~0.7 s and ~10 s of audio, as inference.py:50-56
            for frames in (63, 862):
                hc = torch.randn(1, 80, frames, device=dev)
                m.infer(hc, 0.6)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                xs = m.infer(hc, 0.6)
                torch.cuda.synchronize()
                res["waveflow"]["inverse_khz_%d" % xs.numel()] = xs.numel() / (time.perf_counter() - t1) / 1000.0
        del m
    except Exception as e:                                    # noqa: BLE001
        res["waveflow"] = {"error": repr(e)}
    try:
        torch.manual_seed(0)
        m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False)
        with torch.no_grad():
            for blk in m.WNs:
                blk.F.end.weight.normal_(0.0, 0.02)
        m = m.to(dev)
        crit = cm.WaveGlowLoss(1.0)
        x = torch.rand(12, 8192, device=dev) * 2 - 1
        c = (torch.rand(12, 4096, device=dev) * 2 - 1) * 0.9

        from constant_memory_waveglow_amd.parallel import FlowTrainer
        trw = FlowTrainer(m, 1.0)                  # the bucketed trainer (per-flow gradient buckets + the embedding-table bucket), re-packs every step
        dt = timed(lambda: trw.step(x, c.clone())[0])
        gfin = bool(torch.isfinite(trw.fg.flat).all())
        res["wsrglow"] = {"workload": "WSRGlow 2x, batch 12 x 8192, fwd+NLL+bwd (FlowTrainer)", "ms_per_step": dt * 1e3,
                          "samples_per_s": 12 * 8192 / dt, "loss": checked(last["loss"]), "grads_finite": gfin,
                          "logged": trw.metrics_dict()}
        del m, trw
    except Exception as e:                                    # noqa: BLE001
        res["wsrglow"] = {"error": repr(e)}
    try:
        # configs/waveglow_LJ_speech_fast.json: the headline network with memory_efficient=False.  Timed at the headline's batch
        # (24 x 16000) so the two lines compare directly; the stored WN activations take ~1.9 GB per flow at that size.
        from constant_memory_waveglow_amd.parallel import FlowTrainer
        torch.manual_seed(0)
        m = build_model(dev)
        m.mem_efficient = False
        tr = FlowTrainer(m, SIGMA)
        x = torch.rand(24, SEG, device=dev) * 2 - 1
        h = torch.randn(24, C2["n_mels"], FRAMES, device=dev)
        dt = timed(lambda: tr.step(x, h)[0])
        res["waveglow_memory_efficient_false"] = {
            "loss": checked(last["loss"]),
            "workload": "WaveGlow 256ch 12 flows (waveglow_LJ_speech_fast.json), batch 24 x 16000, fwd+NLL+bwd from stored activations",
            "ms_per_step": dt * 1e3, "samples_per_s": 24 * SEG / dt,
            "workspace_gb": sum(b.numel() for b in m._engine.buffers._ws.values()) / 1e9}
        del m, tr
    except Exception as e:                                    # noqa: BLE001
        res["waveglow_memory_efficient_false"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return res


KCLASS = {0: "conv store / residual / data-gradient (EPI_STORE)", 1: "gate conv (EPI_GATE)", 2: "residual + skip conv (EPI_RESSKIP)",
          3: "gate backward (EPI_DGATE)", 4: "weight gradient", 5: "layer launch: gate conv + residual product",
          6: "rank-2ic skip path: end conv from partial rows / gate planes, P = G gate^T"}
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def kernel_rooflines(trainer, x, h, split, steps=2, top=12, step_fn=None, box_tflops=None):
    """Per-SHAPE rooflines of the conv / weight-gradient launches of the training step: every such launch of `steps` extra steps is
    bracketed with HIP events on the launch stream (wg_timer_create(-1, ..): all classes) and reported with the shape the library
    attaches to it (wg_timer_read_info): class, M x K, columns, algorithmic HBM bytes.  Grouped by (class, M, K); per group: launches per
    step, average duration, algorithmic TFLOP/s against the matrix peak (bf16 2.5 PF with three issued products per fp32 product = 833 TF
    algorithmic; 157.3 TF in the exact-fp32 mode) and algorithmic GB/s against 8 TB/s; `bound` = whichever floor is higher."""
    from constant_memory_waveglow_amd import _lib
    L = _lib.lib()
    cap = 4096 * steps
    t = L.wg_timer_create(-1, cap)
    torch.cuda.synchronize()
    L.wg_timer_attach(t)
    for _ in range(steps):
        if step_fn is not None:
            step_fn()
        else:
            trainer.step(x, h)
    torch.cuda.synchronize()
    L.wg_timer_attach(None)
    n = L.wg_timer_count(t)
    ms = (C.c_float * n)()
    info = (C.c_longlong * (5 * n))()
    L.wg_timer_read(t, ms, n)
    L.wg_timer_read_info(t, info, n)
    L.wg_timer_destroy(t)
    ms = np.frombuffer(ms, dtype=np.float32)
    info = np.frombuffer(info, dtype=np.int64).reshape(n, 5)
    peak_tf = BF16_MFMA_PEAK_TFLOPS / 3.0 if split else FP32_MFMA_PEAK_TFLOPS
    groups = {}
    for d, (cls, M, K, cols, by) in zip(ms, info):
        groups.setdefault((int(cls), int(M), int(K)), []).append((float(d), int(cols), int(by)))
    total_ms = float(ms.sum()) / steps
    rows = []
    for (cls, M, K), v in groups.items():
        us = 1e3 * sum(a for a, _, _ in v) / len(v)
        cols = v[0][1]
        flop = 2.0 * M * K * cols
        by = float(np.mean([b for _, _, b in v]))
        tf, gbs = flop / (us * 1e-6) / 1e12, by / (us * 1e-6) / 1e9
        f_m, f_h = tf / peak_tf, gbs / HBM_PEAK_GBS
        rows.append({"kernel": KCLASS.get(cls, str(cls)), "M": M, "K": K, "columns": cols, "launches_per_step": len(v) / steps,
                     "avg_us": us, "ms_per_step": us * len(v) / steps / 1e3, "flop_per_launch": flop, "bytes_per_launch": by,
                     "tflops_algorithmic": tf, "gbs_algorithmic": gbs, "bound": "mfma" if f_m >= f_h else "hbm", "frac": max(f_m, f_h),
                     "frac_mfma": f_m, "frac_hbm": f_h,
                     **({"frac_of_box": 3.0 * tf / box_tflops} if (box_tflops and split) else {})})
    rows.sort(key=lambda r: -r["ms_per_step"])
    return {"timed_ms_per_step": total_ms, "mfma_peak_tflops_algorithmic": peak_tf, "hbm_peak_gbs": HBM_PEAK_GBS,
            "note": "HIP events around every conv / weight-gradient launch of %d extra steps (the events cost a few per cent; the headline "
                    "step is timed without them); bytes = every operand plane once + weights + epilogue planes" % steps,
            "kernels": rows[:top]}


def f32_mode(dev, x, h):
    """The same step with the contractions on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32; WG_PRECISION=f32): the number that needs no
    argument about operand splitting, next to the bf16x3 headline.  1 warm-up + 5 timed steps; the gate kernel against the fp32 roof."""
    from constant_memory_waveglow_amd import _lib
    from constant_memory_waveglow_amd.parallel import FlowTrainer
    old = os.environ.get("WG_PRECISION")
    os.environ["WG_PRECISION"] = "f32"
    try:
        model = build_model(dev)
        tr = FlowTrainer(model, SIGMA)
        L = _lib.lib()
        tr.step(x, h)
        torch.cuda.synchronize()
        steps = 5
        timer = L.wg_timer_create(_lib.K_CONV_GATE, 2 * C2["flows"] * C2["depth"] * steps)
        L.wg_timer_attach(timer)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _, _ = tr.step(x, h)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        L.wg_timer_attach(None)
        n = L.wg_timer_count(timer)
        buf = (C.c_float * n)()
        L.wg_timer_read(timer, buf, n)
        L.wg_timer_destroy(timer)
        gate_ms = float(np.mean(np.frombuffer(buf, dtype=np.float32)))
        B = x.shape[0]
        gate_flop = 2.0 * (C2["radix"] * C2["residual_channels"] + C2["n_mels"]) * 2 * C2["dilation_channels"] * B * (SEG // C2["n_group"])
        ach = gate_flop / (gate_ms * 1e-3) / 1e12
        return {"dtype": "f32 (v_mfma_f32_32x32x2_f32, bit-exact fp32 fma chains)", "ms_per_step": dt * 1e3, "value": B * SEG / dt,
                "loss": float(loss), "steps": steps,
                "roofline": {"kernel": "convgemm_kernel<EPI_GATE>", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach / FP32_MFMA_PEAK_TFLOPS, "launch_ms": gate_ms}}
    finally:
        if old is None:
            os.environ.pop("WG_PRECISION", None)
        else:
            os.environ["WG_PRECISION"] = old
        torch.cuda.empty_cache()



def make_workload(name, dev, batch, rank):
    """The three benchmarked training steps behind one interface (BASELINE.json configs[1..4]):
      waveglow  configs/waveglow_LJ_speech.json   batch 24 x 16000, sigma 0.7   FlowTrainer (wg_train_step, per-flow gradient buckets)
      waveflow  configs/waveflow_LJ_speech.json   batch 12 x 16000, sigma 0.7   autograd (_WaveFlowFn) + ONE 24 MB gradient collective
      wsrglow   configs/wsrglow_vctk_2x.json      batch 12 x 8192,  sigma 1.0   FlowTrainer (+ the embedding-table bucket)
    step() = forward + NLL + backward to every parameter gradient (+ the mean all-reduce when the process group has > 1 rank or the
    collectives are forced); weights re-packed every step, inputs resident in HBM."""
    import constant_memory_waveglow_amd as cm
    from constant_memory_waveglow_amd.parallel import FlowTrainer, GradSync
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    if name == "waveglow":
        B = batch or 24
        model = build_model(dev)
        trainer = FlowTrainer(model, SIGMA)
        x = torch.rand(B, SEG, device=dev, generator=g) * 2 - 1          # as tests/test_fwd_bwd.py:28 upstream
        h = torch.randn(B, C2["n_mels"], FRAMES, device=dev, generator=g)
        kcat = C2["radix"] * C2["residual_channels"] + C2["n_mels"]
        return dict(model=model, trainer=trainer, sync=trainer.sync, batch=B, segment=SEG, x=x, h=h, generator=g,
                    step=lambda: trainer.step(x, h)[0], metric="audio samples/sec (fwd+bwd) WaveGlow-256ch seg=16000",
                    workload="WaveGlow 256ch, 12 flows, seg=16000, batch=%d per GPU (waveglow_LJ_speech.json), forward + NLL + "
                             "constant-memory backward" % B,
                    step_flop_per_sample=STEP_FLOP_PER_SAMPLE, gate_launches_per_step=2 * C2["flows"] * C2["depth"],
                    gate_flop_per_launch=2.0 * kcat * 2 * C2["dilation_channels"] * B * (SEG // C2["n_group"]),
                    gate_what="dilated k=3 conv + mel conditioning + gate", profile_prefix="")
    if name == "wsrglow":
        B = batch or 12
        torch.manual_seed(0)
        model = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False)
        with torch.no_grad():
            for blk in model.WNs:
                blk.F.end.weight.normal_(0.0, 0.02)
        model = model.to(dev)
        trainer = FlowTrainer(model, 1.0)
        seg = 8192
        x = torch.rand(B, seg, device=dev, generator=g) * 2 - 1
        c = (torch.rand(B, seg // 2, device=dev, generator=g) * 2 - 1) * 0.9
        # per time step of 16 samples and flow: 8 layers x 2 x (3 x 256 + 3659) x 512 (conv + conditioning) + W_o + start / end
        kcat = 3 * 256 + 3659
        fwd = (8 * (2 * kcat * 512 + 2 * 256 * 512) - 2 * 256 * 256 + 2 * 256 * 16) * 12 / 16.0
        return dict(model=model, trainer=trainer, sync=trainer.sync, batch=B, segment=seg, x=x, h=c, generator=g,
                    step=lambda: trainer.step(x, c.clone())[0], metric="audio samples/sec (fwd+bwd) WSRGlow 2x seg=8192",
                    workload="WSRGlow 2x (wsrglow_vctk_2x.json: 12 flows, n_group 16, 3659 conditioning channels, 229.7 M parameters), "
                             "seg=8192, batch=%d per GPU, conditioning front-end + forward + NLL + constant-memory backward" % B,
                    step_flop_per_sample=3.0 * fwd, gate_launches_per_step=2 * 12 * 8,
                    gate_flop_per_launch=2.0 * kcat * 512 * B * (seg // 16), gate_what="dilated k=3 conv + 3659-channel conditioning + gate",
                    profile_prefix="_wsr_")
    B = batch or 12
    torch.manual_seed(0)
    model = cm.WaveFlow(flows=8, n_group=64, n_mels=80, use_conv1x1=False, memory_efficient=False, dilation_channels=64,
                        residual_channels=64, skip_channels=64, bias=False)
    with torch.no_grad():
        for wn in model.WNs:
            wn.end.weight.normal_(0.0, 0.02)
    model = model.to(dev)
    crit = cm.WaveGlowLoss(SIGMA)
    sync = GradSync()
    params = list(model.parameters())
    sync.broadcast_params(params)
    x = torch.rand(B, SEG, device=dev, generator=g) * 2 - 1
    h = torch.randn(B, 80, FRAMES, device=dev, generator=g)

    def step():
        model.zero_grad(set_to_none=True)
        model._engine.packed.key = None            # as in training, where the weights change: re-pack every step
        z, ld = model(x, h)
        loss = crit(z, ld)
        loss.backward()
        sync.all_reduce_params(params)
        return loss
    # per (height row, time) position and layer: 3x3 conv 2 x 64 x 128 x 9, conditioning 2 x 80 x 128, W_o 2 x 64 x 128; 8 flows x 8 layers;
    # 63 of an item's 64 rows are WN2D inputs (waveflow.py:196-206)
    fwd = 8 * 8 * (2 * 64 * 128 * 9 + 2 * 80 * 128 + 2 * 64 * 128) * 63 / 64.0
    return dict(model=model, trainer=None, sync=sync, batch=B, segment=SEG, x=x, h=h, generator=g, step=step, n_param_bytes=4 * sum(p.numel() for p in params),
                metric="audio samples/sec (fwd+bwd) WaveFlow-64ch seg=16000",
                workload="WaveFlow 64ch, 8 flows, n_group 64 (waveflow_LJ_speech.json), seg=16000, batch=%d per GPU, forward + NLL + backward" % B,
                step_flop_per_sample=3.0 * fwd, gate_launches_per_step=2 * 8 * 8,
                gate_flop_per_launch=2.0 * (9 * 64 + 80) * 128 * B * 64 * (SEG // 64), gate_what="3x3 dilated conv + mel conditioning + gate",
                profile_prefix="_wf_")


def comm_report(wl, args, use_dist, world, rank, dev, dt_own, ms_per_step, barrier):
    """What the N > 1 line needs for a first real multi-GPU run to be diagnosed: bytes and buckets of the gradient exchange, every
    rank's own step time, and `exposed_ms` = the step minus the SAME step with the collectives skipped (all ranks skip together, so
    nothing waits), i.e. the communication time the backward did not hide."""
    sync, tr = wl["sync"], wl["trainer"]
    if tr is not None:
        buckets = [4 * tr.fg.comm_slice(b).numel() for b in range(tr.n_buckets)]
    else:
        buckets = [wl["n_param_bytes"]]
    per_rank = torch.zeros(world, device=dev, dtype=torch.float64)
    per_rank[rank] = dt_own / args.steps * 1e3
    dist.all_reduce(per_rank)
    k = max(1, min(args.steps, 5))
    sync.skip = True
    try:
        wl["step"]()
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            wl["step"]()
        barrier()
        t_nc = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    finally:
        sync.skip = False
    dist.all_reduce(t_nc, op=dist.ReduceOp.MAX)
    nocomm_ms = float(t_nc.item()) / k * 1e3
    # a ring all-reduce moves 2 (N - 1) / N of the buffer per rank over its xGMI links
    return {"backend": dist.get_backend(), "bytes_per_step": int(sum(buckets)), "buckets": len(buckets),
            "bucket_mb": [round(b / 1e6, 2) for b in buckets], "order": "backward order: last flow first, then the upsampler" if tr is not None else "one collective behind the backward",
            "per_rank_ms": [round(float(v), 3) for v in per_rank.tolist()], "ms_per_step": ms_per_step,
            "ms_per_step_collectives_skipped": nocomm_ms, "exposed_ms": ms_per_step - nocomm_ms, "steps_skipped_run": k,
            "wire_bytes_per_rank_ring": int(2.0 * (world - 1) / max(world, 1) * sum(buckets))}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", choices=("waveglow", "waveflow", "wsrglow"), default="waveglow",
                    help="waveglow: BASELINE.json configs[1] / [2] (the headline); waveflow: configs[3] (waveflow_LJ_speech.json); "
                         "wsrglow: configs[4] (wsrglow_vctk_2x.json) -- same launcher, same JSON contract")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the model's config -- 24 / 12 / 12)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-inverse", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the WSRGlow / WaveFlow step timings (SURVEY.md 8f rows)")
    ap.add_argument("--no-box", action="store_true", help="skip the 0.5 s box-calibration probe (`box` in the line)")
    ap.add_argument("--spawn", action="store_true", help="go through the launcher even for --gpus 1 (a 1-rank RCCL group)")
    ap.add_argument("--dry-run", action="store_true", help="launcher only: print the child command and environment as JSON, start nothing")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="HARNESS TEST ONLY: more ranks than GPUs (rank r on GPU r %% n_gpus) over a gloo group -- RCCL refuses two ranks "
                         "on one device; the line is marked and its value means nothing for scaling")
    ap.add_argument("--selftest", choices=("ok", "fail"), default=None,
                    help="launcher / rank plumbing without a GPU: the ranks form a gloo group, reduce a number and rank 0 prints a JSON "
                         "line ('fail': rank 1 exits with code 3 instead) -- what tests/test_bench_launcher_cpu.py runs")
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_plan(args, argv, port=None):
    """(command, environment additions) of the child that runs the N ranks.  The reference gets its ranks from Lightning
    (`pl.Trainer(gpus=N, strategy=DDPPlugin(...))`, train.py:73-78: one process per GPU); here torch.distributed.run starts them, with
    the rendezvous on 127.0.0.1 (the container's hostname may not resolve).  `--spawn` / `--dry-run` are the launcher's own flags."""
    own = {"--spawn", "--dry-run"}
    rest = [a for a in argv if a not in own]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else _free_port()),
           os.path.abspath(__file__)] + rest
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0",       # the host driver only supports dmabuf IPC (RCCL across processes)
           "WG_BENCH_LAUNCHED": "1",                 # the ranks use the process group even when there is only one of them
           "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, args.gpus))))}
    return cmd, env


def launch(args, argv):
    """Parent side of `bench.py --gpus N`: never touches the GPU (torch.cuda.device_count() does not initialise it on this image),
    starts the ranks as a CHILD process -- not an exec --, relays their output to stderr and rank 0's JSON line to stdout."""
    import subprocess
    cmd, extra = launcher_plan(args, argv)
    if args.dry_run:
        print(json.dumps({"launcher": True, "cmd": cmd, "env": extra, "n_ranks": args.gpus}))
        return 0
    if not args.selftest and not args.oversubscribe:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print("bench.py: --gpus %d but this node shows %d GPU(s)" % (args.gpus, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.update(extra)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True, bufsize=1)
    line = None
    for raw in proc.stdout:                       # rank 0 prints the JSON line; anything else the ranks say goes to stderr
        t = raw.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(raw)
    rc = proc.wait()
    if rc != 0:
        print("bench.py: the %d-rank child exited with code %d" % (args.gpus, rc), file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 1
    sys.stderr.flush()
    print(line, flush=True)
    return 0


def selftest_rank(args, world, rank):
    """The rank side of the plumbing on CPU (gloo): rendezvous from the environment torch.distributed.run sets, a MAX reduction like the
    one the timing uses, rank 0's line last on stdout."""
    dist.init_process_group("gloo")
    if args.selftest == "fail" and rank == 1:
        os._exit(3)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "selftest", "model": args.model, "value": float(t.item()), "n_gpus": world, "rccl_ranks": world,
                          "steps": args.steps, "warmup": args.warmup, "launched": os.environ.get("WG_BENCH_LAUNCHED") == "1"}), flush=True)
    dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    in_rank = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not in_rank and (args.gpus > 1 or args.spawn or args.dry_run):
        return launch(args, argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # n_gpus in the line is the real world size; a mismatch with --gpus is a harness error, not something to paper over
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    if args.selftest:
        return selftest_rank(args, world, rank)
    # a group of ONE rank still runs the collectives when it was launched (or WG_BENCH_FORCE_DIST=1): the RCCL path on one GPU
    use_dist = world > 1 or os.environ.get("WG_BENCH_FORCE_DIST") == "1" or os.environ.get("WG_BENCH_LAUNCHED") == "1"
    if use_dist:
        os.environ["WG_BENCH_FORCE_DIST"] = "1" if world == 1 else os.environ.get("WG_BENCH_FORCE_DIST", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.oversubscribe:
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from constant_memory_waveglow_amd import _lib

    if args.gpus > 1 and not args.oversubscribe and not (use_dist and dist.get_backend() == "nccl" and dist.get_world_size() == args.gpus):
        print("bench.py: --gpus %d needs %d RCCL ranks (got backend %s, %d ranks)"
              % (args.gpus, args.gpus, dist.get_backend() if use_dist else None, dist.get_world_size() if use_dist else 0), file=sys.stderr)
        return 2
    wl = make_workload(args.model, dev, args.batch, rank)
    model, trainer, B, SEGW = wl["model"], wl["trainer"], wl["batch"], wl["segment"]
    x, h = wl["x"], wl["h"]
    g = wl["generator"]
    step = wl["step"]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    L = _lib.lib()
    per_step_gate = wl["gate_launches_per_step"]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):                                   # the headline: NOTHING attached, no event beyond the two brackets
        loss = step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0                             # this rank's own time (before the barrier: comm.per_rank_ms)
    barrier()
    dt_all = time.perf_counter() - t0
    logged = trainer.metrics_dict() if (rank == 0 and trainer is not None) else None     # (of the last TIMED step: before comm_report's un-reduced ones)
    tmax = torch.tensor([dt_all], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * SEGW * args.steps / dt
    comm = comm_report(wl, args, use_dist, world, rank, dev, dt_own, ms_per_step, barrier) if use_dist else None
    # (behind the timed steps: half a second of full-power matrix work on ONE rank in front of a max-reduced multi-rank headline would
    # heat that rank's GPU alone; the other ranks wait at the barrier below)
    box = box_probe(dev) if rank == 0 and not args.no_box else None
    # the dominant kernel's launch duration: HIP events around every launch of its class over `args.steps` EXTRA steps (every rank runs
    # them: the collectives need all ranks), outside the timed region; the library reports which instantiation each launch ran
    # (every class is timed: the gate conv runs on its own -- class K_CONV_GATE -- or, at the training shapes since round 5, inside the
    # one-launch layer -- K_LAYER: gate conv + residual product; the roofline describes whichever takes more of the step)
    timer = L.wg_timer_create(-1, 4096 * args.steps) if rank == 0 else None
    barrier()
    if timer:
        L.wg_timer_attach(timer)
    for _ in range(args.steps):
        step()
    barrier()
    if timer:
        L.wg_timer_attach(None)

    out = None
    if rank == 0:
        n_all = L.wg_timer_count(timer)
        buf = (C.c_float * n_all)()
        info = (C.c_longlong * (5 * n_all))()
        L.wg_timer_read(timer, buf, n_all)
        L.wg_timer_read_info(timer, info, n_all)
        ms_all = np.frombuffer(buf, dtype=np.float32)
        info = np.frombuffer(info, dtype=np.int64).reshape(n_all, 5)
        per_class = {c: float(ms_all[info[:, 0] == c].sum()) for c in (_lib.K_CONV_GATE, _lib.K_LAYER)}
        dom = max(per_class, key=per_class.get)
        sel = np.nonzero(info[:, 0] == dom)[0]
        # (a class holds launches of more than one K since the first layer's conv reads xa through the composed weight -- DESIGN.md 4g: its
        # gate conv is a few chunks long.  The roofline line is about the full-K launches: the others would be credited FLOPs they do not do)
        if sel.size:
            sel = sel[info[sel, 2] == info[sel, 2].max()]
        names = {}
        nb = C.create_string_buffer(256)
        for i in sel:
            if L.wg_timer_read_name(timer, int(i), nb, 256) >= 0:
                names[nb.value.decode()] = names.get(nb.value.decode(), 0) + 1
        L.wg_timer_destroy(timer)
        n = int(sel.size)
        gate_ms = float(np.mean(ms_all[sel])) if n else float("nan")
        # algorithmic FLOPs of one launch: the gate conv's, or (layer launch) gate conv + residual product: 2 M K columns of what the
        # library attaches to the launch (wg_timer_read_info: for the layer K counts the residual's channels scaled to the gate's rows)
        gate_flop = wl["gate_flop_per_launch"] if dom == _lib.K_CONV_GATE or not n else float(2.0 * np.median(info[sel, 1] * info[sel, 2] * info[sel, 3]))
        what = wl["gate_what"] if dom == _lib.K_CONV_GATE else wl["gate_what"] + ", then W_o's residual rows on the same workgroup (one launch per layer)"
        achieved = gate_flop / (gate_ms * 1e-3) / 1e12
        split = _lib.default_precision() != _lib.PREC_F32
        # bf16x3: every fp32 product costs three bf16 MFMAs; the roofline is the bf16 matrix pipe and only the
        # algorithmic FLOPs are credited (the 3x is overhead, not work) -- SURVEY.md 8d
        peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        # the instantiation most of the timed launches ran (a model may take more than one: small flows, other tile forms)
        site = max(names, key=names.get) if names else ""
        kprefix = " + ".join(launch_site_to_rocprof(p) for p in site.split(" + ")) if site else ""
        traffic, traffic_src, kfull = site_traffic(site, wl["profile_prefix"]) if site else (None, None, None)
        if traffic_src:
            traffic_src += " (committed rocprofv3 --pmc summary of this command; not re-measured in this run)"
        gate_alone = None
        galone = np.nonzero(info[:, 0] == _lib.K_CONV_GATE)[0]
        if galone.size:
            galone = galone[info[galone, 2] == info[galone, 2].max()]
        if dom == _lib.K_LAYER and galone.size:                          # the gate conv launches that run on their own (a WN's last layer)
            g_ms = float(np.mean(ms_all[galone]))
            g_ach = wl["gate_flop_per_launch"] / (g_ms * 1e-3) / 1e12
            gate_alone = {"launch_ms": g_ms, "launches_timed": int(galone.size), "flop_per_launch": wl["gate_flop_per_launch"], "achieved": g_ach,
                          "frac": g_ach / peak}
        STEP_FLOP = wl["step_flop_per_sample"]
        out = {
            "metric": wl["metric"],
            "value": value, "unit": "samples/s", "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if use_dist and not args.oversubscribe else 0,
            **({"oversubscribed": "HARNESS TEST: %d gloo ranks share %d GPU(s); not a scaling number" % (world, torch.cuda.device_count())}
               if args.oversubscribe else {}),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (contractions as split bf16x3 MFMA, fp32 accumulate)" if split else "f32", "data": "synthetic",
            "config": {"workload": wl["workload"] + (" + RCCL grad all-reduce" if world > 1 else ""),
                       "global_batch": B * world, "segment": SEGW, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "kernel": "%s (EPI_GATE: %s)" % (kfull or " + ".join(p + ">" if "<" in p else p for p in kprefix.split(" + ")), what),
                         "kernel_launch_sites": names,
                         "class_ms_timed": {"gate conv on its own": per_class[_lib.K_CONV_GATE], "layer launch": per_class[_lib.K_LAYER]},
                         **({"gate_conv_alone": gate_alone} if gate_alone else {}),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": gate_ms, "launches_timed": n, "launch_timing": "HIP events around every launch of the class over %d extra steps "
                                                                                    "behind the timed ones (nothing is attached while the headline runs)" % args.steps,
                         "flop_per_launch": gate_flop,
                         "mfma_tflops_issued": achieved * (3 if split else 1),
                         # issued TFLOP/s of the dominant launch over what THIS GPU sustains on the library's fixed matrix-pipe + LDS loop
                         # without global traffic (`box`): separates a slow box from a slow kernel in one number
                         **({"frac_of_box": achieved * (3 if split else 1) / box["tflops_issued"]} if (box and box.get("tflops_issued") and split) else {}),
                         "x_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS},
            "step_flop_per_sample": STEP_FLOP,
            "step_tflops_algorithmic": value * STEP_FLOP / 1e12 / world,
            "step_frac_of_fp32_mfma_peak": value * STEP_FLOP / 1e12 / world / FP32_MFMA_PEAK_TFLOPS,
            "step_frac_of_bf16_mfma_peak": value * STEP_FLOP / 1e12 / world / BF16_MFMA_PEAK_TFLOPS,
            "loss": float(loss),
        }
        if box is not None:
            out["box"] = box
        if logged is not None:
            out["logged"] = logged                     # the scalars LightModel.training_step logs (lightning.py:58-64), rank-mean, of the last timed step
        if comm is not None:
            out["comm"] = comm
        if args.model != "waveglow":
            wl["no_inverse"] = args.no_inverse
            args.no_inverse = args.no_extra = args.no_cpu = True         # the secondary legs belong to the headline workload
            if world == 1:
                try:
                    out["roofline"]["kernels"] = kernel_rooflines(trainer, x, h, split, step_fn=step if trainer is None else None,
                                                                      box_tflops=(box or {}).get("tflops_issued"))
                except Exception as e:                                    # noqa: BLE001
                    out["roofline"]["kernels"] = {"error": repr(e)}
            if args.model == "waveflow" and world == 1 and not wl.get("no_inverse"):
                with torch.no_grad():                                     # synthesis (row-by-row inverse), as inference.py:50-56
                    for frames in (63, 862):
                        hc = torch.randn(1, 80, frames, device=dev)
                        model.infer(hc, 0.6)
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        xs = model.infer(hc, 0.6)
                        torch.cuda.synchronize()
                        out["inverse_khz_%d" % xs.numel()] = xs.numel() / (time.perf_counter() - t1) / 1000.0
        if not args.no_inverse:
            with torch.no_grad():
                for frames in (63, 862):                                  # 16 128 samples and ~10 s of audio
                    hc = torch.randn(1, C2["n_mels"], frames, device=dev, generator=g)
                    model.infer(hc, 0.6)
                    costs = []
                    for _ in range(5):                                    # one call between two synchronisations, as inference.py:50-56
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        xs = model.infer(hc, 0.6)
                        torch.cuda.synchronize()
                        costs.append(time.perf_counter() - t1)
                    out["inverse_khz_%d" % xs.numel()] = xs.numel() / sorted(costs)[2] / 1000.0
                    t1 = time.perf_counter()
                    for _ in range(10):                                   # a queue of utterances: the host runs ahead of the device
                        xs = model.infer(hc, 0.6)
                    torch.cuda.synchronize()
                    out["inverse_khz_%d_queued" % xs.numel()] = xs.numel() * 10 / (time.perf_counter() - t1) / 1000.0
            out["inverse_khz"] = out["inverse_khz_%d" % (862 * 256)]
            # throughput next to latency: the reference's infer takes h[B, ...] (base.py:42-55) -- eight 0.7 s utterances in one call
            hb = torch.randn(8, C2["n_mels"], 63, device=dev, generator=g)
            with torch.no_grad():
                model.infer(hb, 0.6)
                costs = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    xs = model.infer(hb, 0.6)
                    torch.cuda.synchronize()
                    costs.append(time.perf_counter() - t1)
            out["inverse_khz_batch8x16128"] = xs.numel() / sorted(costs)[2] / 1000.0
            # the reference's published configuration (musicnet: 18 flows, early outputs every 6, WN depth 4, hop 512): 10 s of audio
            mus = build_model(dev, cfg=MUSICNET)
            hm = torch.randn(1, MUSICNET["n_mels"], 431, device=dev, generator=g)
            with torch.no_grad():
                mus.infer(hm, 0.6)
                costs = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    xm = mus.infer(hm, 0.6)
                    torch.cuda.synchronize()
                    costs.append(time.perf_counter() - t1)
            mus_khz = xm.numel() / sorted(costs)[2] / 1000.0
            out["inverse_khz_musicnet_%d" % xm.numel()] = mus_khz
            del mus
            # synthesis is the forward's FLOPs (13.38 MFLOP per sample) on the same matrix pipe: 2.5 PF bf16, three issued per product
            peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
            out["inverse_roofline"] = {
                "bound": "mfma", "unit": "TFLOP/s", "peak": peak, "flop_per_sample": FWD_FLOP_PER_SAMPLE,
                "cases": {**{str(nn): {"khz": out["inverse_khz_%d" % nn], "achieved": out["inverse_khz_%d" % nn] * 1e3 * FWD_FLOP_PER_SAMPLE / 1e12,
                                       "frac": out["inverse_khz_%d" % nn] * 1e3 * FWD_FLOP_PER_SAMPLE / 1e12 / peak} for nn in (63 * 256, 862 * 256)},
                          "8x16128": {"khz": out["inverse_khz_batch8x16128"], "achieved": out["inverse_khz_batch8x16128"] * 1e3 * FWD_FLOP_PER_SAMPLE / 1e12,
                                      "frac": out["inverse_khz_batch8x16128"] * 1e3 * FWD_FLOP_PER_SAMPLE / 1e12 / peak}},
                "musicnet": {"config": "configs/musicnet_config.json: 18 flows, n_early_every 6, WN depth 4, hop 512, 80 mels; %d samples" % xm.numel(),
                             "khz": mus_khz, "flop_per_sample": fwd_flop_per_sample(MUSICNET),
                             "achieved": mus_khz * 1e3 * fwd_flop_per_sample(MUSICNET) / 1e12,
                             "frac": mus_khz * 1e3 * fwd_flop_per_sample(MUSICNET) / 1e12 / peak,
                             "reference_published_khz": MUSICNET_PUBLISHED_KHZ,
                             "reference_published_on": "GTX 1080 Ti (README.md:64-67 of the reference: 'around 470kHz'); other hardware, quoted for scale",
                             "x_reference_published": mus_khz / MUSICNET_PUBLISHED_KHZ},
                "note": "one call between two synchronisations as inference.py:50-56; a 16 128-sample utterance is a chain of ~250 small "
                        "launches (64 x 64 tiles on every CU, DESIGN.md section 4a iii), the 10 s utterance fills the chip"}
        if world == 1 and args.model == "waveglow":
            # outside the metric (SURVEY.md 8d excludes the optimizer): one Adam step over all 53.66 M parameters on the flat buffers
            from constant_memory_waveglow_amd.parallel import FlatAdam
            opt = FlatAdam(trainer, lr=1e-4)
            trainer.optimizer = None                          # timed stand-alone; trainer.step would run it per bucket
            opt.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                opt.step()
            torch.cuda.synchronize()
            out["adam_step_ms"] = (time.perf_counter() - t1) / 5 * 1e3
        if world == 1 and args.model == "waveglow":                   # (extra steps on one rank only would wait for collectives forever)
            try:
                out["roofline"]["kernels"] = kernel_rooflines(trainer, x, h, split, box_tflops=(box or {}).get("tflops_issued"))
            except Exception as e:                                    # noqa: BLE001 -- diagnostics never touch the headline line
                out["roofline"]["kernels"] = {"error": repr(e)}
        if world == 1 and not args.no_extra:
            out["f32_mode"] = f32_mode(dev, x, h)
            out["other_models"] = other_models(dev)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which would otherwise be flushed
        # behind it at exit
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
