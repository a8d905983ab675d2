# usage: bash tools/experiments/pair_fetch.sh v1 v2 ...: FETCH_SIZE (KiB, raw) and duration of the paired weight-gradient launch per variant
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  WGFLOW_LIB=$R/variants/lib_$v.so rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf_$v -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-inverse --no-extra > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/pf_$v/*/*counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "pair" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("$v", "launches", len(v), "FETCH_SIZE avg %.0f MiB"%(sum(v)/len(v)/1024))
PY
done
