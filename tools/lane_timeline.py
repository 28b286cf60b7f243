#!/usr/bin/env python3
"""Per-lane timeline of a two-lane bench run from a rocprofv3 --kernel-trace CSV: for every step of every lane (= HIP queue) the
duration of the geometry before the UNet pass, of the UNet pass, of the geometry after it, and the gaps between consecutive UNet passes
of the device.  usage: tools/lane_timeline.py <kernel_trace.csv> [first_step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
t0 = min(e[0] for e in ev)
ms = lambda t: (t - t0) / 1e6
lanes = sorted({e[3] for e in ev if "k_enc0_pp" in e[2]})
passes = []
for q in lanes:
    L = sorted(e for e in ev if e[3] == q)
    first = [e for e in L if "k_obb_face_area2" in e[2]]
    scale = [e for e in L if "k_anp_rows" in e[2]]      # the last kernel of the chain in front of the pass (the 16-bit network scales the image where it reads it)
    enc = [e for e in L if "k_enc0_pp" in e[2]]
    head = [e for e in L if "k_dec0b_head_pp" in e[2]]
    app = [e for e in L if "k_apply_csys" in e[2]]
    n = min(len(first), len(scale), len(enc), len(head), len(app))
    pre = [round(ms(scale[i][1]) - ms(first[i][0]), 2) for i in range(skip, n)]
    wait = [round(ms(enc[i][0]) - ms(scale[i][1]), 2) for i in range(skip, n)]
    un = [round(ms(head[i][1]) - ms(enc[i][0]), 2) for i in range(skip, n)]
    post = [round(ms(app[i][1]) - ms(head[i][1]), 2) for i in range(skip, n)]
    gap = [round(ms(first[i + 1][0]) - ms(app[i][1]), 2) for i in range(skip, n - 1)]
    print(f"lane {q}: geometry before UNet {pre}\n        wait for the UNet turn {wait}\n        UNet pass {un}\n        geometry after {post}\n        host gap to the next step {gap}")
    passes += [(enc[i][0], head[i][1]) for i in range(skip, n)]
passes.sort()
print("gaps between consecutive UNet passes of the device (ms):", [round((passes[i + 1][0] - passes[i][1]) / 1e6, 2) for i in range(len(passes) - 1)])
span = (passes[-1][1] - passes[0][0]) / 1e6
print(f"UNet passes cover {sum(b - a for a, b in passes) / 1e6 / span:.3f} of the span; {span / len(passes):.2f} ms per pass incl. gaps")
