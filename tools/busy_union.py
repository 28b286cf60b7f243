"""Device occupancy over time from a rocprofv3 kernel trace: python tools/busy_union.py kernel_trace.csv [lo hi]
Looks at the window [lo, hi] (fractions of the trace span, default 0.35 0.75 = steady state of a bench run): share of the
window with 0 / 1 / 2+ kernels in flight, and per kernel the time attributed to it (an instant shared by n kernels
counts 1/n for each)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.75)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
t0, t1 = min(e[0] for e in ev), max(e[1] for e in ev)
a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
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
