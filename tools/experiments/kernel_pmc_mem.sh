# usage: bash tools/experiments/kernel_pmc_mem.sh <tag>: memory-path counters (L2 hits / misses / tag stalls, L1 pending / FIFO stalls, texture-addresser
# stalls (NOT collected: a pass with TA_* counters aborts rocprofv3 on this image and hangs the call), fabric read requests and their queue level) of the conv / weight-gradient kernels over tools/kbench.py, one small counter set per pass
tag=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcm_${tag}_$i -- python3 $R/tools/kbench.py --iters 1 > $R/gpurun_out/pmcm_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, 8):
    fs = glob.glob("$R/gpurun_out/pmcm_${tag}_%d/*/*counter_collection.csv" % i)
    if not fs:
        print(i, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "wgrad16t" in k or "convgemm16q_kernel<5, 2, 2" in k or "convgemm16q_kernel<4, 2, 1" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(i, k[:44], {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, "n=%d" % max(len(v) for v in acc[k].values()))
PY
