#!/usr/bin/env python3
"""Which kernels of the other lane's chain slow a UNet pass down?  From a two-lane rocprofv3 --kernel-trace CSV: every UNet kernel
instance's extra time over a per-position baseline (the same launch with nothing beside it: tools/probes/grid224_trace.sh) is
regressed (non-negative least squares) on the time it shared with each kernel name of the other queues.
usage: tools/interference.py <two_lane_trace.csv> <baseline_trace.csv> [first_pass]"""
import csv, sys, statistics, collections
import numpy as np
from scipy.optimize import nnls

def load(path):
    rows = list(csv.DictReader(open(path)))
    return sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sh::", "").replace("sh::", "")[:44], r["Queue_Id"]) for r in rows)

is_head = lambda n: "k_dec0b_head_pp" in n
def passes(ev, skip):
    out = []
    for q in sorted({e[3] for e in ev if "k_enc0_pp" in e[2]}):
        L = [e for e in ev if e[3] == q]
        i = n = 0
        while i < len(L):
            if "k_enc0_pp" not in L[i][2]:
                i += 1; continue
            j = i
            while j < len(L) and not is_head(L[j][2]): j += 1
            if j >= len(L): break
            n += 1
            if n > skip: out.append(L[i:j + 1])
            i = j + 1
    return out

skip = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ev2, ev1 = load(sys.argv[1]), load(sys.argv[2])
base = collections.defaultdict(list)
for P in passes(ev1, skip):
    for k, e in enumerate(P): base[k].append(e[1] - e[0])
base = {k: statistics.median(v) for k, v in base.items()}
unet_q = {e[3] for e in ev2 if "k_enc0_pp" in e[2]}
names = sorted({e[2] for e in ev2 if "conv" not in e[2] and "enc0" not in e[2] and "dec0" not in e[2] and "pack" not in e[2]})
col = {n: i for i, n in enumerate(names)}
others = [e for e in ev2 if e[2] in col]
starts = np.array([e[0] for e in others])
A, y = [], []
for P in passes(ev2, skip):
    for k, e in enumerate(P):
        row = np.zeros(len(names))
        lo = np.searchsorted(starts, e[0] - 5_000_000); hi = np.searchsorted(starts, e[1])
        for o in others[lo:hi]:
            ov = min(e[1], o[1]) - max(e[0], o[0])
            if ov > 0: row[col[o[2]]] += ov
        A.append(row / 1e3); y.append((e[1] - e[0] - base[k]) / 1e3)
A, y = np.array(A), np.array(y)
x, _ = nnls(A, y)
share = A.sum(0) * x
print(f"UNet kernel instances {len(y)}; extra time per pass {y.sum() / (len(y) / len(base)):.0f} us; explained {share.sum() / (len(y) / len(base)):.0f} us")
for i in np.argsort(-share)[:18]:
    if share[i] > 0: print(f"  {names[i]:44s} us of UNet time lost per us shared {x[i]:5.2f}   per pass {share[i] / (len(y) / len(base)):6.0f} us  (shared {A[:, i].sum() / (len(y) / len(base)):6.0f} us per pass)")
