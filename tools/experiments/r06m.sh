# full GPU suite (summary line kept) + kernel statistics of 3 training steps
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/${1:-r06m}_tests.log 2>&1
grep -E "passed|failed|FAILED|rror" gpurun_out/${1:-r06m}_tests.log | tail -5
bash tools/experiments/prof_stats.sh ${1:-r06m} > /dev/null 2>&1
st=$(ls -t $(find gpurun_out/prof_${1:-r06m} -name "*kernel_stats.csv") | head -1); cp $st gpurun_out/${1:-r06m}_kernel_stats.csv; rm -rf gpurun_out/prof_${1:-r06m}
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${1:-r06m}_kernel_stats.csv')))
for r in rows[:14]:
    if 'box_probe' in r['Name']: continue
    print('%-70s %5s x %9.1f us = %8.2f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
