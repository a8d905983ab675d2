cd $GRAFT_REPO_ROOT
{
bash tools/experiments/run_variants.sh g0 gnostore gsc1 gnt g0 gsc1
cd /tmp && export TMPDIR=/tmp
WGFLOW_LIB=$GRAFT_REPO_ROOT/variants/lib_g0.so rocprofv3 --kernel-trace --stats -d /tmp/prof_g0 -o g0 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --iters 3 --fwd-only > /tmp/prof.log 2>&1
f=$(find /tmp/prof_g0 -name "*kernel_trace.csv" | head -1); echo $f
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
out=[]
for r in rows:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.append((r['Kernel_Name'][:60], (en-st)/1e3, (st-prev)/1e3 if prev else 0.0))
    prev=en
for o in out[-45:]: print("%-60s dur %8.1f us  gap %6.1f us" % o)
PY
} > gpurun_out/r05e.txt 2>&1
cat gpurun_out/r05e.txt
