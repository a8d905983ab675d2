# usage: bash tools/experiments/inv_ab.sh v1 v2 ...: synthesis rates (WaveGlow 0.7 s / 10 s utterance, kHz) per variant library
for v in "$@"; do
  printf "$v: "
  WGFLOW_LIB=variants/lib_$v.so python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('inverse %.0f kHz (16128)  %.0f kHz (220672)  queued %.0f / %.0f' % (d['inverse_khz_16128'], d['inverse_khz_220672'], d['inverse_khz_16128_queued'], d['inverse_khz_220672_queued']))"
done
