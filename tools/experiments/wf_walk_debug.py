"""bisecting aid for the row walk (wg_stage.h): one small WaveFlow inverse, prints whether it survived and the error against the launch path"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fill
import constant_memory_waveglow_amd as cm
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "wf8"
cfg = fill.WF_CONFIGS[name]
B, N, F = fill.WF_SHAPES[name]
specs = fill.waveflow_param_specs(cfg)
P = fill.fill_params(specs, name + "/")
audio, mel = fill.waveflow_inputs(name, B, N, F, cfg["n_mels"])
m = cm.WaveFlow(use_conv1x1=False, memory_efficient=False, bias=False, **cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
m = m.to(dev)
g = np.load(os.path.join(ROOT, "tests", "golden", "model_%s.npz" % name))
print("keys", [k for k in g.files if "inv" in k or k in ("z",)], flush=True)
with torch.no_grad():
    x, ld = m.reverse(torch.from_numpy(g["z"]).to(dev), torch.from_numpy(mel).to(dev))
torch.cuda.synchronize()
print("survived; |x - audio| max", float(np.abs(x.cpu().numpy() - audio).max()), "logdet", ld.cpu().numpy(), flush=True)
