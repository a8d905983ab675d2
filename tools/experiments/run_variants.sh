for v in base rot swap halfb halfab noepi nostore base; do echo "=== $v"; WGFLOW_LIB=variants/lib_$v.so python tools/kbench.py --iters 3 2>&1 | grep -v Warning | tail -7; done
