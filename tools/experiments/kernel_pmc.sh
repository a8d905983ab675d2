# usage: bash tools/experiments/kernel_pmc.sh <tag> [variant]: PMC passes over tools/kbench.py (one coupling forward + backward at the C2 shape) ->
# gpurun_out/pmc_<tag>_{fetch,write,mfma,lds}/...counter_collection.csv ; with a variant name the library variants/lib_<variant>.so is used
tag=${1:-x}
R=$GRAFT_REPO_ROOT
if [ -n "$2" ]; then export WGFLOW_LIB=$R/variants/lib_$2.so; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_fetch -- python3 $R/tools/kbench.py --iters 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_write -- python3 $R/tools/kbench.py --iters 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_mfma -- python3 $R/tools/kbench.py --iters 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_${tag}_lds -- python3 $R/tools/kbench.py --iters 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for kind in ("fetch", "write", "mfma", "lds"):
    fs = glob.glob("$R/gpurun_out/pmc_${tag}_%s/*/*counter_collection.csv" % kind)
    if not fs:
        print(kind, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "wgrad" in k or "convgemm16q" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(kind, k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, "n=%d" % max(len(v) for v in acc[k].values()))
PY
