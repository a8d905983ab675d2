# usage: bash tools/experiments/ab_bench.sh name1 name2 ...  -> ms per training step (bench.py, 10 steps) per variant, interleaved twice
for rep in 1 2; do for v in "$@"; do printf "%s " $v; WGFLOW_LIB=variants/lib_$v.so python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --no-inverse --no-box 2>/dev/null | python tools/experiments/bench_rows.py | head -${ROWS:-1}; done; done
