# usage: bash tools/experiments/fetch_ab.sh v1 v2 ...: FETCH_SIZE (HBM read KiB, raw counter) and duration per conv kernel and variant library, one training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  WGFLOW_LIB=$R/variants/lib_$v.so rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch_$v -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-inverse --no-extra > /dev/null 2>&1
  echo "== $v"
  python3 - "$R/gpurun_out/prof_fetch_$v" <<'P'
import csv,glob,os,sys,collections
d=sys.argv[1]
cc=sorted(glob.glob(d+"/*/*_counter_collection.csv"),key=os.path.getmtime)[-1]; kt=sorted(glob.glob(d+"/*/*_kernel_trace.csv"),key=os.path.getmtime)[-1]
dur={r["Dispatch_Id"]:(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(kt))}
acc=collections.defaultdict(lambda:[0,0.0,0.0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"]!="FETCH_SIZE" or "convgemm16q" not in r["Kernel_Name"]: continue
    t=dur.get(r["Dispatch_Id"],0)
    name=r["Kernel_Name"].split("(")[0].replace("void ","")
    if "<0, 2, 1>" in name:
        name+= " k256" if 30<t<50 else " dh" if 88<t<118 else " skip" if 118<t<160 else " dy" if 170<t<240 else " other"
    a=acc[name]; a[0]+=1; a[1]+=float(r["Counter_Value"]); a[2]+=t
for k,(n,f,t) in sorted(acc.items()):
    print("  %-40s n=%4d  fetch %.0f MB/launch (x2)  %.1f us"%(k,n,2*f/n*1024/1e6,t/n))
P
done
