#!/usr/bin/env python3
"""Gaps between consecutive kernels INSIDE the UNet passes of a bench run (rocprofv3 --kernel-trace CSV): per boundary (kernel -> next
kernel of the same queue, from k_enc0_pp to the head kernel) the median idle time, and per pass the sum of kernel durations against
the span of the pass.  usage: tools/unet_gaps.py <kernel_trace.csv> [first_pass]"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
short = lambda n: n.split("(")[0].replace("void sh::", "").replace("sh::", "")[:40]
is_head = lambda n: "k_dec0b_head_pp" in n
gaps, sums, spans, durs = {}, [], [], {}
for q in sorted({e[3] for e in ev if "k_enc0_pp" in e[2]}):
    L = [e for e in ev if e[3] == q]
    i, npass = 0, 0
    while i < len(L):
        if "k_enc0_pp" not in L[i][2]:
            i += 1
            continue
        j = i
        while j < len(L) and not is_head(L[j][2]):
            j += 1
        if j >= len(L):
            break
        npass += 1
        if npass > skip:
            P = L[i:j + 1]
            sums.append(sum(e[1] - e[0] for e in P) / 1e6)
            spans.append((P[-1][1] - P[0][0]) / 1e6)
            for k, e in enumerate(P):
                durs.setdefault((k, short(e[2])), []).append((e[1] - e[0]) / 1e3)
            for a, b in zip(P, P[1:]):
                gaps.setdefault((short(a[2]), short(b[2])), []).append((b[0] - a[1]) / 1e3)
        i = j + 1
print(f"passes {len(sums)}: kernels {statistics.median(sums):.3f} ms, span {statistics.median(spans):.3f} ms, idle inside a pass {statistics.median(spans) - statistics.median(sums):.3f} ms")
for k, v in sorted(gaps.items(), key=lambda kv: -statistics.median(kv[1])):
    print(f"  {k[0]:40s} -> {k[1]:40s} median {statistics.median(v):7.1f} us  (n {len(v)})")
print("kernel durations by position in the pass (median us):")
for k, v in sorted(durs.items()):
    print(f"  {k[0]:2d} {k[1]:40s} {statistics.median(v):8.1f}")
