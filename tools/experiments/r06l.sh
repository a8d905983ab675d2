cd $GRAFT_REPO_ROOT
WGFLOW_LIB=variants/lib_unit16.so python -m pytest tests/test_gpu_parity.py -x -q -k "wide_batch or c2_single or flattened or test_full_size_properties" 2>&1 | grep -E "passed|failed" 
ROWS=4 bash tools/experiments/ab_bench.sh base unit16
