"""How the host of the GPU box scales the torch-CPU oracle: topology / quota, then C1 steps with W worker processes x T threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fill
from oracle import torch_cpu
n, quota, firsts = torch_cpu.host_cpu_budget()
print("logical CPUs allowed %d, cgroup quota %s CPUs, physical cores allowed %d; loadavg %s" % (n, quota, len(firsts), open("/proc/loadavg").read().strip()))
for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "GOMP_CPU_AFFINITY", "KMP_AFFINITY", "MKL_NUM_THREADS"):
    print(k, os.environ.get(k))
cfg = fill.CONFIGS["c1"]; specs = fill.model_param_specs(cfg); tab = fill.table(specs, fill.fill_params(specs, "c1/"))
audio, h = fill.inputs("c1", 2, 4000, 16, cfg["n_mels"])
for W, T in ((1, 8), (2, 8), (4, 8), (8, 8), (16, 8), (8, 16), (16, 4), (32, 4), (32, 2), (64, 2), (64, 1), (128, 1)):
    if W * T > len(firsts):
        continue
    t0 = time.time()
    r = torch_cpu.time_parallel(cfg, tab, audio[:1], h[:1], fill.SIGMA, workers=W, threads=T, runs=3)
    print("W %3d x T %2d: %8.0f samples/s, step median fastest %.3f slowest %.3f s  (%.1f s)" % (W, T, r["samples_per_s"], r["step_s_median_fastest_worker"], r["step_s_median_slowest_worker"], time.time() - t0), flush=True)
