"""Condenses rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python tools/profile_summary.py <round-tag> <kernel_stats.csv> [<fetch_counter.csv> <write_counter.csv>]

Writes profiles/<tag>_kernel_stats.csv (our kernels only, torch RNG/fill kernels dropped, names shortened) and, when the
two PMC passes are given, profiles/<tag>_hbm_traffic.json with per-kernel HBM bytes per launch:
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream --
MI355X_MICROARCH.md, "HBM" -- hence the factor 2; the two counters need separate passes: TCC has 4 slots)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name.strip()


def ours(name):
    return not ("at::native" in name or "rocclr" in name)


def pmc_summary(tag, paths):
    """python tools/profile_summary.py --pmc <tag> <counter_collection.csv>...: per-kernel averages of every counter in the given
    rocprofv3 --pmc passes -> profiles/<tag>_pmc.json, plus the derived shares (matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /
    (4 SIMDs x SQ_BUSY_CU_CYCLES); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)."""
    sums, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for path in paths:
        for r in csv.DictReader(open(path)):
            if ours(r["Kernel_Name"]):
                k = short(r["Kernel_Name"])
                sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
    res = {}
    for k in sorted(sums):
        e = {c: round(sums[k][c] / cnt[k][c], 1) for c in sorted(sums[k])}
        e["launches"] = max(cnt[k].values())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CU_CYCLES"):
            e["mfma_busy_share"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * e["SQ_BUSY_CU_CYCLES"]), 4)
        if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = round(e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"], 4)
        res[k] = e
    out = os.path.join(ROOT, "profiles", "%s_pmc.json" % tag)
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/kbench.py --iters 1 (one coupling forward + backward at the C2 "
                         "shape; separate passes per counter group); per-launch averages over the launches of each kernel",
               "kernels": res}, open(out, "w"), indent=1)
    print("wrote", out)


def shape_summary(tag, bench_json):
    """python tools/profile_summary.py --shapes <tag> <bench line .json>: rocprofv3's kernel statistics lump every product that runs on one
    kernel instantiation into one row (convgemm16q_kernel<0, 2, 1> is the residual conv, the data-gradient conv, the skip product, the
    conditioning gradient and the start / end convs); the per-SHAPE table comes from the bench line, whose `roofline.kernels` object times
    every conv / weight-gradient launch with HIP events and groups them by the (class, M, K) the library attaches to the launch
    (wg_timer_read_info).  Written as profiles/<tag>_shape_rooflines.csv."""
    d = json.loads(open(bench_json).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    out = os.path.join(ROOT, "profiles", "%s_shape_rooflines.csv" % tag)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel_class", "M", "K", "columns", "launches_per_step", "avg_us", "ms_per_step", "gflop_per_launch", "mb_per_launch",
                    "tflops_algorithmic", "gbs_algorithmic", "bound", "frac_of_bound", "frac_mfma(peak %.0f TF)" % k["mfma_peak_tflops_algorithmic"],
                    "frac_hbm(peak %.0f GB/s)" % k["hbm_peak_gbs"]])
        for r in k["kernels"]:
            w.writerow([r["kernel"], r["M"], r["K"], r["columns"], "%.1f" % r["launches_per_step"], "%.1f" % r["avg_us"], "%.2f" % r["ms_per_step"],
                        "%.2f" % (r["flop_per_launch"] / 1e9), "%.1f" % (r["bytes_per_launch"] / 1e6), "%.1f" % r["tflops_algorithmic"],
                        "%.0f" % r["gbs_algorithmic"], r["bound"], "%.3f" % r["frac"], "%.3f" % r["frac_mfma"], "%.3f" % r["frac_hbm"]])
    print("wrote", out)


def main():
    if sys.argv[1] == "--pmc":
        return pmc_summary(sys.argv[2], sys.argv[3:])
    if sys.argv[1] == "--shapes":
        return shape_summary(sys.argv[2], sys.argv[3])
    tag, stats = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct_of_gpu_time", "min_us", "max_us"])
        for r in rows:
            if ours(r["Name"]):
                w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                            "%.2f" % (float(r["AverageNs"]) / 1e3), "%.2f" % (100 * float(r["TotalDurationNs"]) / total),
                            "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3)])
    print("wrote", out)
    if len(sys.argv) >= 5:
        acc = {}
        for key, path in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            sums, cnt = defaultdict(float), defaultdict(int)
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == key and ours(r["Kernel_Name"]):
                    k = short(r["Kernel_Name"])
                    sums[k] += float(r["Counter_Value"])
                    cnt[k] += 1
            acc[key] = {k: (sums[k] / cnt[k], cnt[k]) for k in sums}
        res = {}
        for k in sorted(acc["FETCH_SIZE"]):
            f, n = acc["FETCH_SIZE"][k]
            wv = acc["WRITE_SIZE"].get(k, (0.0, 0))[0]
            res[k] = {"launches": n, "fetch_kib_raw": f, "write_kib": wv, "hbm_bytes_per_launch": (2 * f + wv) * 1024}
        out = os.path.join(ROOT, "profiles", "%s_hbm_traffic.json" % tag)
        json.dump({"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, averaged over launches", "kernels": res}, open(out, "w"), indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()
