#!/usr/bin/env python3
"""Registers, scratch and LDS of the kernels whose name contains a pattern, from the code-object metadata of the built library
(a spill in a ping-pong / LDS-DMA kernel breaks its counted waits: check after every edit).
  python tools/kernel_regs.py <pattern> [library]"""
import os, re, struct, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shoulder_amd", "lib", "libshoulder_hip.so")
data = open(path, "rb").read()
for o in [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)]:
    shoff = struct.unpack_from("<Q", data, o + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", data, o + 0x3A)
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[o:o + shoff + shentsize * shnum]); name = f.name
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", name], capture_output=True, text=True).stdout
    os.unlink(name)
    for blk in notes.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        sym = g("name")
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        if pat in dem:
            print(f"{dem[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} spills {g('vgpr_spill_count')} lds {g('group_segment_fixed_size')}")
