# usage: bash tools/experiments/small_launch_ab.sh v1 v2 ...: single-utterance WaveGlow synthesis (bench.py's inverse figures) and WaveFlow's row-by-row inverse per library variant
for v in "$@"; do
  echo "== $v"
  WGFLOW_LIB=variants/lib_$v.so python tools/experiments/infer_latency.py 2>/dev/null | grep "single call"
  WGFLOW_LIB=variants/lib_$v.so python tools/experiments/wf_inverse_long.py 2>/dev/null | grep samples
done
