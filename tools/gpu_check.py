"""Stage-by-stage parity printout of the HIP engine against the CPU oracle (developer tool, run via gpurun)."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fill                                   # noqa: E402
from oracle import wg_oracle as orc           # noqa: E402
import constant_memory_waveglow_amd as cm     # noqa: E402
from make_golden import COUPLING_CASES        # noqa: E402

dev = torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-30))


def amax(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return float(np.abs(a - b).max())


def stage(fn):
    t0 = time.time()
    try:
        fn()
    except Exception:
        traceback.print_exc()
    torch.cuda.synchronize()
    print("   [%s %.1fs]" % (fn.__name__, time.time() - t0), flush=True)


def invconv():
    for c in (2, 4, 8):
        W = fill.orthogonal("chk/W%d" % c, c)
        x = fill.uniform("chk/x%d" % c, (3, c, 200))
        gz = fill.normal("chk/gz%d" % c, (3, c, 200))
        for rev in (False, True):
            blk = cm.InvertibleConv1x1(c, memory_efficient=True).to(dev)
            blk.weight.data.copy_(T(W).unsqueeze(-1))
            xt = T(x).requires_grad_(True)
            xin = xt.clone()
            y, ld = blk.reverse(xin) if rev else blk(xin)
            freed = xin.untyped_storage().size() == 0
            ((y * T(gz)).sum() + ld * 0.37).backward()
            yo, ldo = (orc.invconv_reverse if rev else orc.invconv_forward)(W, x)
            xr, dxo, dWo = orc.invconv_backward(W, yo, gz, 0.37, reverse=rev)
            print("invconv c=%d rev=%d: y %.2e logdet %.2e dx %.2e dW %.2e freed=%s rebuilt %.2e" % (
                c, rev, amax(y, yo), abs(float(ld) - float(ldo)), rel(xt.grad, dxo), rel(blk.weight.grad[:, :, 0], dWo),
                freed, amax(xin, x)))


def coupling():
    for cname, cs in COUPLING_CASES.items():
        tag = "coupling/" + cname
        wn = dict(in_channels=cs["c"] // 2, aux_channels=cs["aux"], residual_channels=cs["wn"], dilation_channels=cs["wn"],
                  skip_channels=cs["wn"], depth=cs["depth"], radix=3)
        specs = fill.wn_param_specs("F.", cs["c"] // 2, cs["aux"], cs["wn"], cs["wn"], cs["wn"], cs["depth"], 3)
        P = fill.fill_params(specs, tag + "/")
        tab = fill.table(specs, P)
        x = fill.uniform(tag + "/x", (cs["B"], cs["c"], cs["T"]))
        y = fill.normal(tag + "/y", (cs["B"], cs["aux"], cs["T"]))
        gz = fill.normal(tag + "/gz", x.shape)
        gls = fill.normal(tag + "/gls", (cs["B"], cs["c"] // 2, cs["T"]))
        for rev in (False, True):
            blk = cm.AffineCouplingBlock(cm.WN, True, zero_init=False, **wn)
            blk.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
            blk = blk.to(dev)
            xt, yt = T(x).requires_grad_(True), T(y).requires_grad_(True)
            xin = xt.clone()
            z, ls = blk.reverse(xin, yt) if rev else blk(xin, yt)
            ((z * T(gz)).sum() + (ls * T(gls)).sum()).backward()
            zo, lso = orc.coupling_apply(wn, tab, x, y, reverse=rev)
            r = orc.coupling_backward(wn, tab, zo, y, gz, gls, reverse=rev)
            named = dict(blk.named_parameters())
            worst, wname = 0.0, ""
            for (n, _, _), g in zip(specs, r["grads"]):
                e = rel(named[n].grad, g)
                if e > worst:
                    worst, wname = e, n
            print("coupling %s rev=%d: z %.2e log_s %.2e dx %.2e dy %.2e rebuilt %.2e worst grad %.2e (%s)" % (
                cname, rev, amax(z, zo), amax(ls, lso), rel(xt.grad, r["dx"]), rel(yt.grad, r["dy"]), amax(xin, x), worst, wname))
            if not rev:
                with torch.no_grad():
                    a, b = blk.F(T(x[:, :cs["c"] // 2]), T(y))
                print("   WN alone: log_s %.2e" % amax(a, lso))


def model(name, B=None, check_grads=True):
    cfg = fill.CONFIGS[name]
    B0, N, F = fill.SHAPES[name]
    B = B or B0
    specs = fill.model_param_specs(cfg)
    P = fill.fill_params(specs, name + "/")
    tab = fill.table(specs, P)
    audio, h = fill.inputs(name, B, N, F, cfg["n_mels"])
    oc = orc.make_config(**cfg)
    t0 = time.time()
    ref = orc.train_step(oc, tab, audio, h, fill.SIGMA, need_dh=True)
    t_or = time.time() - t0
    m = cm.WaveGlow(memory_efficient=True, bias=False, **cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m = m.to(dev)
    crit = cm.WaveGlowLoss(fill.SIGMA)
    x, ht = T(audio), T(h).requires_grad_(True)
    yup = m._upsample_h(ht.detach())
    yo = orc.upsample(oc, P["upsampler.bias"], P["upsampler.weight_g"], P["upsampler.weight_v"], h, yup.shape[2])
    print("model %s B=%d: upsample %.2e" % (name, B, amax(yup, yo)))
    z, ld = m(x, ht)
    loss = crit(z, ld)
    print("   z %.2e  logdet %.2e (|ld| %.1f)  loss %.2e" % (amax(z, ref["z"]), amax(ld, ref["logdet"]), float(np.abs(ref["logdet"]).max()),
                                                             abs(float(loss) - ref["loss"])), flush=True)
    if check_grads:
        loss.backward()
        named = dict(m.named_parameters())
        worst, wname, bad = 0.0, "", 0
        for (n, _, _), g in zip(specs, ref["grads"]):
            e = rel(named[n].grad, g)
            if e > 1e-4:
                bad += 1
                if bad <= 12:
                    print("      BAD %s rel %.2e" % (n, e))
            if e > worst:
                worst, wname = e, n
        print("   grads: worst rel %.2e (%s), %d/%d above 1e-4; dh %.2e" % (worst, wname, bad, len(specs), rel(ht.grad, ref["dh"])))
    with torch.no_grad():
        xr, ldr = m.reverse(z.detach(), ht.detach())
    xo, ldo = orc.inverse(oc, tab, ref["z"], h)
    print("   inverse vs oracle %.2e  roundtrip %.2e  logdet_fwd+rev %.2e  oracle step %.1fs" % (
        amax(xr, xo), amax(xr, audio), float((ld.detach() + ldr).abs().max()), t_or), flush=True)


def micro():
    model("micro")


def c1():
    model("c1")


def c2():
    model("c2")


if __name__ == "__main__":
    which = sys.argv[1:] or ["invconv", "coupling", "micro", "c1"]
    for w in which:
        stage(globals()[w])
