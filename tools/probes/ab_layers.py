import sys, json, glob, collections
# mean per-layer ms per arm over rounds: python ab_layers.py tagA tagB
arms = sys.argv[1:]
res = {}
for a in arms:
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(f"gpurun_out/ab/pp_{a}_r*.log")):
        L = [json.loads(l) for l in open(f) if l.startswith('{"layers_ms"')]
        if len(L) < 2: continue
        for k, v in L[-1]["layers_ms"].items(): acc[k].append(v)
    res[a] = {k: sum(v) / len(v) for k, v in acc.items()}
keys = list(res[arms[0]].keys())
print("layer".ljust(14) + "".join(a[:14].rjust(16) for a in arms))
for k in keys: print(k.ljust(14) + "".join(f"{res[a].get(k, float('nan')):16.4f}" for a in arms))
print("total".ljust(14) + "".join(f"{sum(res[a].values()):16.4f}" for a in arms))
