"""Compiles csrc/wgflow.hip for gfx950 into csrc/libwgflow.so (in-tree, next to the sources).

    python constant-memory-waveglow_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libwgflow.so")


def _sources():
    """everything the one translation unit includes: csrc/*.hip, csrc/*.h and the public ABI header"""
    own = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    return own + [os.path.join("..", "..", "include", "wgflow.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in _sources())


def build(force=False, verbose=False, defines=(), out=None):
    """defines / out: developer A/B builds (tools/kbench.py), e.g. build(True, defines=["WG_OPT_X=1"], out="/tmp/x.so")"""
    if out is None and not force and not needs_build():
        return OUT
    out = out or OUT
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed"] + ["-D" + d for d in defines] + ["-o", out, os.path.join(CSRC, "wgflow.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
