#!/usr/bin/env python3
"""What the built library carries: code objects (one per translation unit), kernels, bytes, matrix / LDS-DMA instruction counts.
  python tools/lib_census.py [shoulder_amd/lib/libshoulder_hip.so]
The fat binary sections of the .so hold clang offload bundles; every gfx950 code object in them is an ELF that llvm-readelf /
llvm-objdump read directly (DESIGN.md section 10 quotes this script's output)."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shoulder_amd", "lib", "libshoulder_hip.so")
data = open(path, "rb").read()
# code objects: ELF images for amdgcn embedded in the bundle ("\x7fELF" + 64-bit little endian + OS/ABI 0x40 AMDGPU_HSA)
offs = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)]
print(f"{path}: {len(data)} bytes, {len(offs)} gfx950 code object(s)")
tot_k = 0
for i, o in enumerate(offs):
    # ELF size: section header offset + count * size
    import struct
    shoff = struct.unpack_from("<Q", data, o + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", data, o + 0x3A)
    size = shoff + shentsize * shnum
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[o:o + size]); name = f.name
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", name], capture_output=True, text=True).stdout
    kernels = sorted({l.split()[-1] for l in syms.splitlines() if " FUNC " in l and (" GLOBAL " in l or " WEAK " in l) and not l.split()[-1].endswith(".kd")})
    kd = sorted({l.split()[-1] for l in syms.splitlines() if l.strip().endswith(".kd")})
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", name], capture_output=True, text=True).stdout
    counts = {k: len(re.findall(k, dis)) for k in ("v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x4_f32", "global_load_lds_dwordx4", "scratch_")}
    print(f"  code object {i}: {size} bytes, {len(kd)} kernels; " + ", ".join(f"{v} {k}" for k, v in counts.items()))
    tot_k += len(kd)
    os.unlink(name)
print(f"  total: {tot_k} kernels")
