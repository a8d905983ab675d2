"""debug: WSRGlow at the timed size vs the reference summary (tests/golden/model_wsr_full.npz): where z differs, under the env given"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fill
import constant_memory_waveglow_amd as cm
dev = torch.device("cuda:0")
name = "wsr_full"
cfg = fill.CONFIGS[name]; B, N, F = fill.SHAPES[name]
specs = fill.model_param_specs(cfg)
P = fill.fill_params(specs, name + "/"); P.update(fill.wsr_tables(name + "/"))
m = cm.WSRGlow(upsample_rate=2, memory_efficient=True, bias=False)
sd = {k: torch.from_numpy(v) for k, v in P.items()}; sd["window"] = torch.hann_window(16)
m.load_state_dict(sd); m = m.to(dev)
audio, c = fill.wsr_inputs(name, B, N, 2)
gold = np.load(os.path.join(ROOT, "tests", "golden", "model_%s.npz" % name))
with torch.no_grad():
    z, ld = m(torch.from_numpy(audio).to(dev), torch.from_numpy(c.copy()).to(dev))
zz = z.cpu().numpy()
eh, et = np.abs(zz[:, :256] - gold["z_head"]), np.abs(zz[:, -256:] - gold["z_tail"])
print(os.environ.get("WG_G192_SPLITK"), os.environ.get("WG_G192"), os.environ.get("WG_PRECISION"), "head", eh.max(1), "tail", et.max(1))
print("tail err by time column (16 samples each), item of max:", et[et.max(1).argmax()].reshape(16, 16).max(1))
print("logdet err", np.abs(ld.cpu().numpy() - gold["logdet"]))
# which quantiser decisions differ from the reference's (indices recorded in the golden)
from constant_memory_waveglow_amd import engine
with torch.no_grad():
    cond = engine.wsr_cond(torch.from_numpy(c.copy()).to(dev), m.mu_enc[1].weight, m.angle_embed.embed.weight).cpu().numpy()
mu_t, ang_t = P["mu_enc.1.weight"], P["angle_embed.embed.weight"]
mu_idx, ang_idx = gold["mu_idx"].astype(np.int64), gold["ang_idx"].astype(np.int64)
L = c.shape[1]; Fr = L // 8
ref_mu = mu_t[mu_idx].reshape(B, Fr, 3200).transpose(0, 2, 1)            # [B, 3200, F]
bad = np.argwhere((cond[:, :3200] != ref_mu).any(1))
print("mu-law: (item, frame) with another decision:", bad.tolist())
ref_ang = ang_t[ang_idx].transpose(0, 1, 3, 2).reshape(B, 450, Fr)
bad2 = np.argwhere((cond[:, 3209:] != ref_ang).any(1))
print("angle: (item, frame) with another decision:", bad2.tolist())
