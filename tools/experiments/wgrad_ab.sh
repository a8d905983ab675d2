# usage: bash tools/experiments/wgrad_ab.sh v1 v2 ...: the weight-gradient launch of one coupling block (tools/kbench.py) per variant, twice
for rep in 1 2 3; do for v in "$@"; do printf "%s: " $v; WGFLOW_LIB=variants/lib_$v.so python tools/kbench.py --iters 3 2>&1 | grep "^wgrad\|whole" | tr '\n' ' '; echo; done; done
