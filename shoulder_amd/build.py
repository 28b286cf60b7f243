"""Build libshoulder_hip.so (gfx950) in-tree with hipcc.

`python -m shoulder_amd.build` or `shoulder_amd.build.build_lib()`.  Every csrc/*.hip is one
translation unit; a unit is recompiled when it or a header it includes is newer than its object.  hipcc
cross-compiles for gfx950 without a GPU.  -ffp-contract=off: the geometry kernels restate
NumPy expressions operation by operation (DESIGN.md "Numerics"); the UNet kernels use explicit
MFMA / fma builtins, which the flag does not affect.
"""
import concurrent.futures
import glob
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libshoulder_hip.so")
OBJDIR = os.path.join(LIBDIR, "obj")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
                  + glob.glob(os.path.join(ROOT, "include", "*.h")))


_INC = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)


def _deps(path, seen=None):
    """`path` and every project header it includes (recursively; searched beside the file, in csrc/ and in include/)."""
    seen = seen if seen is not None else set()
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        text = f.read()
    for inc in _INC.findall(text):
        for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, "include")):
            cand = os.path.normpath(os.path.join(base, inc))
            if os.path.exists(cand):
                _deps(cand, seen)
                break
    return seen


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc", "-Wall", "-Wmisleading-indentation",
            "-Wno-unused-function"] + os.environ.get("SHOULDER_HIPCC_FLAGS", "").split()      # (the variable: experiments only)


def _obj(src):
    return os.path.join(OBJDIR, os.path.splitext(os.path.basename(src))[0] + ".o")


def _obj_stale(src):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in _deps(src)) or os.path.getmtime(__file__) > t


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build_lib(force=False, verbose=True):
    """One object per translation unit (csrc/*.hip), compiled in parallel when stale, then one link: a change to one kernel family
    recompiles its unit only."""
    if not force and not is_stale():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    units = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    todo = [s for s in units if force or _obj_stale(s)]

    def compile_one(src):
        cmd = [_hipcc()] + _flags() + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(compile_one, todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", "-o", LIB] + [_obj(s) for s in units]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
