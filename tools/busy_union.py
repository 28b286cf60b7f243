"""Device occupancy over time from a rocprofv3 kernel trace: python tools/busy_union.py kernel_trace.csv [lo hi]
Looks at the window [lo, hi] (fractions of the trace span; default: the steady state of a bench run, found from the
per-step k_obb_candidates launches): share of the
window with 0 / 1 / 2+ kernels in flight, and per kernel the time attributed to it (an instant shared by n kernels
counts 1/n for each)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
t0, t1 = min(e[0] for e in ev), max(e[1] for e in ev)
if len(sys.argv) > 3:
    lo, hi = float(sys.argv[2]), float(sys.argv[3])
    a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
else:
    # steady state of a bench run: from the 8th step's first kernel to the 6th-last step's (one k_obb_candidates launch per step)
    steps = sorted(s for s, e, n in ev if "k_obb_candidates" in n)
    a, b = steps[min(8, len(steps) - 2)], steps[max(len(steps) - 6, min(8, len(steps) - 2) + 1)]
pts = []
for s, e, n in ev:
    s, e = max(s, a), min(e, b)
    if e > s:
        pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort(key=lambda p: (p[0], p[1]))
active = collections.Counter(); hist = collections.Counter(); attr = collections.Counter(); raw = collections.Counter()
prev = a; nact = 0
for t, d, n in pts:
    dt = t - prev
    if dt > 0:
        hist[min(nact, 3)] += dt
        for k, c in active.items():
            if c > 0:
                attr[k] += dt * c / nact
                raw[k] += dt * c
    prev = t
    active[n] += d; nact += d
hist[min(nact, 3)] += b - prev
W = b - a
print(f"window {W * 1e-6:.1f} ms; idle {hist[0] / W:.3f}, one kernel {hist[1] / W:.3f}, two {hist[2] / W:.3f}, three+ {hist[3] / W:.3f}")
for k, v in attr.most_common(16):
    print(f"{v / W:6.3f} attributed  {raw[k] / W:6.3f} raw  {k[:90]}")
