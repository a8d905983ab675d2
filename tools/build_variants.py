"""Builds A/B variants of libwgflow.so into gpurun_out/variants/ (developer tool)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "constant-memory-waveglow_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
out = os.path.join(ROOT, "variants")
os.makedirs(out, exist_ok=True)
for spec_ in sys.argv[1:]:
    name, _, defs = spec_.partition(":")
    defines = [d for d in defs.split(",") if d]
    print(name, defines, b.build(True, defines=defines, out=os.path.join(out, "lib_%s.so" % name)))
