# quick check of a kernel change: the tests that reach the rank-2ic skip path, then kernel statistics of 3 training steps
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "wide_batch or shape_sweep or flattened or c2_full or full_size_properties or inverse_and_infer or trainer_step or stored_activation or repeated_steps" 2>&1 | tail -4
bash tools/experiments/prof_stats.sh ${1:-r06f} > /dev/null 2>&1
st=$(ls -t $(find gpurun_out/prof_${1:-r06f} -name "*kernel_stats.csv") | head -1); cp $st gpurun_out/${1:-r06f}_kernel_stats.csv; rm -rf gpurun_out/prof_${1:-r06f}
tail -1 gpurun_out/prof_${1:-r06f}.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step (under the profiler)', d['ms_per_step'])"
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${1:-r06f}_kernel_stats.csv')))
for r in rows[:26]:
    if 'box_probe' in r['Name']: continue
    print('%-70s %5s x %9.1f us = %8.2f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
