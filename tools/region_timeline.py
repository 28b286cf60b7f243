#!/usr/bin/env python3
"""Where a bench region's time goes, from a rocprofv3 --kernel-trace CSV: every UNet pass of the device (k_enc0_pp start -> k_dec0b_head_pp
end) with its lane (queue), the idle time of the matrix kernels in front of it, and the chain / tail of its step; then the region as the
driver times it: first geometry kernel of the first timed step -> last kernel.
  usage: tools/region_timeline.py <kernel_trace.csv> <passes in front of the region> <timed steps>      (0 0: every pass of the process)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
warm, steps = int(sys.argv[2]), int(sys.argv[3])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
enc = [e for e in ev if "k_enc0_pp" in e[2]]
head = [e for e in ev if "k_dec0b_head_pp" in e[2]]
first = [e for e in ev if "k_hullpre_extremes" in e[2] or "k_obb_face_area2" in e[2]]
app = [e for e in ev if "k_apply_csys" in e[2]]
n = min(len(enc), len(head))
area = [e for e in ev if "k_obb_face_area2" in e[2]]
if steps <= 0:      # every pass of the process, to find the region by eye
    steps, warm = n, 0
enc, head, area, app = enc[warm:warm + steps], head[warm:warm + steps], area[warm:warm + steps], app[warm:warm + steps]
t0 = area[0][0]
ms = lambda t: (t - t0) / 1e6
print("step lane   chain starts  pass starts   pass ends   pass ms   idle before the pass   tail ends")
prev_end = None
for i in range(steps):
    idle = ms(enc[i][0]) - (ms(prev_end) if prev_end else 0.0)
    print(f"{i:4d} {enc[i][3]:>4s} {ms(area[i][0]):12.2f} {ms(enc[i][0]):12.2f} {ms(head[i][1]):11.2f} {ms(head[i][1]) - ms(enc[i][0]):9.2f} {idle:12.2f} {ms(app[i][1]):19.2f}")
    prev_end = head[i][1]
last = max(app[-1][1], head[-1][1])
print(f"region (first OBB kernel of the first timed step -> last kernel): {ms(last):.2f} ms = {ms(last) / steps:.3f} ms per step; "
      f"UNet passes {sum(ms(head[i][1]) - ms(enc[i][0]) for i in range(steps)):.2f} ms; before the first pass {ms(enc[0][0]):.2f} ms; behind the last {ms(last) - ms(head[-1][1]):.2f} ms")
