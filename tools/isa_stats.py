#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file, block by block.
  python tools/isa_stats.py file.s <kernel-name-substring> [--min N]
Prints per basic block (label): total instructions, MFMA, LDS reads/writes, vector memory, VALU, SALU, waits, and the
register / LDS / scratch footprint from the kernel descriptor: what a wave has to issue between two MFMAs is what bounds the
32-channel kernels."""
import re, sys, collections

def main():
    path, name = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 20
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and name in m.group(1):
            start = i; sym = m.group(1); break
    if start is None:
        print("kernel not found"); return 1
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print(sym)
    blocks = collections.OrderedDict(); cur = "entry"; blocks[cur] = []
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m: cur = m.group(1); blocks[cur] = []
            continue
        if re.match(r"^(\.LBB\d+_\d+):", s): cur = s.split(":")[0]; blocks[cur] = []; continue
        blocks[cur].append(s.split()[0])
    def cls(op):
        if op.startswith("v_mfma"): return "mfma"
        if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_rd"
        if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_wr"
        if op.startswith("ds_"): return "ds_other"
        if op.startswith(("global_load_lds", "buffer_load")) : return "vm_ld"
        if op.startswith(("global_load", "flat_load", "scratch_load")): return "vm_ld"
        if op.startswith(("global_store", "flat_store", "scratch_store", "buffer_store")): return "vm_st"
        if op.startswith(("global_atomic", "flat_atomic")): return "vm_at"
        if op.startswith("s_waitcnt"): return "wait"
        if op.startswith("s_barrier"): return "barrier"
        if op.startswith("s_nop"): return "nop"
        if op.startswith("s_"): return "salu"
        if op.startswith("v_"): return "valu"
        return "other"
    tot = collections.Counter()
    for b, ops in blocks.items():
        c = collections.Counter(cls(o) for o in ops)
        tot.update(c)
        if len(ops) >= minn:
            top = collections.Counter(o for o in ops if cls(o) == "valu").most_common(8)
            print(f"{b:12s} n={len(ops):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
            print("             valu top: " + ", ".join(f"{k}:{v}" for k, v in top))
    print("TOTAL " + " ".join(f"{k}={v}" for k, v in sorted(tot.items())))
    for l in lines[end:end + 80]:
        if any(k in l for k in ("NumVgprs", "NumAgprs", "TotalNumVgprs", "ScratchSize", "LDSByteSize", "Occupancy", "NumSgprs")): print(l.strip())
    return 0

sys.exit(main())
