# usage: bash tools/experiments/run_variants.sh name1 name2 ...   (variants/lib_<name>.so built by tools/build_variants.py)
for v in "$@"; do echo "=== $v"; WGFLOW_LIB=variants/lib_$v.so python tools/kbench.py --iters 3 --fwd-only 2>&1 | grep -v "Warn\|WeightNorm\|amdgpu.ids" | tail -4; done
