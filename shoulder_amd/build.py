"""Build libshoulder_hip.so (gfx950) in-tree with hipcc.

`python -m shoulder_amd.build` or `shoulder_amd.build.build_lib()`.  The library is rebuilt
when any source under shoulder_amd/csrc or include/ is newer than the .so.  hipcc
cross-compiles for gfx950 without a GPU.  -ffp-contract=off: the geometry kernels restate
NumPy expressions operation by operation (DESIGN.md "Numerics"); the UNet kernels use explicit
MFMA / fma builtins, which the flag does not affect.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libshoulder_hip.so")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
                  + glob.glob(os.path.join(ROOT, "include", "*.h")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build_lib(force=False, verbose=True):
    if not force and not is_stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-gpu-rdc", "-Wall", "-Wmisleading-indentation", "-Wno-unused-function", "-o", LIB] + sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    cmd += os.environ.get("SHOULDER_HIPCC_FLAGS", "").split()      # experiments only (e.g. -DSH_... ablation switches)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
