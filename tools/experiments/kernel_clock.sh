# usage: bash tools/experiments/kernel_clock.sh v1 v2 ...: effective clock (GRBM_GUI_ACTIVE / 8 / duration) and matrix-pipe share of the weight-gradient
# launch of tools/kbench.py per library variant (MI355X_MICROARCH.md, "DVFS give-back")
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  WGFLOW_LIB=$R/variants/lib_$v.so rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk_$v -- python3 $R/tools/kbench.py --iters 2 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
d = glob.glob("$R/gpurun_out/clk_$v/*/")[0]
dur = {}
for r in csv.DictReader(open(glob.glob(d + "*kernel_trace.csv")[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(d + "*counter_collection.csv")[0])):
    acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
rows = collections.defaultdict(list)
for k, c in acc.items():
    ns, name = dur[k]
    name = name.split("(")[0].replace("void ", "")
    if ("wgrad16t" in name or "pair" in name or "convgemm16q_kernel<1" in name) and ns > 50000:
        rows[name].append((ns, c["GRBM_GUI_ACTIVE"] / 8 / ns, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_BUSY_CU_CYCLES"])))
for name, v in rows.items():
    n = len(v)
    print("$v %-34s n=%d  %.1f us  clock %.2f GHz  mfma busy %.3f" % (name, n, sum(x[0] for x in v) / n / 1e3, sum(x[1] for x in v) / n, sum(x[2] for x in v) / n))
PY
done
