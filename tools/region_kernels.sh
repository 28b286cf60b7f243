#!/bin/bash
# rocprofv3 kernel durations of the UNet kernels inside the two-lane region and with one lane alone (same box, same process flags):
#   tools/region_kernels.sh <tag>      -> gpurun_out/region_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
for lanes in 2 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rk_${tag}_$lanes -o rk -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra-legs --lanes $lanes > gpurun_out/rk_${tag}_$lanes.json 2> gpurun_out/rk_${tag}_$lanes.err || exit 1
done
python3 - "$tag" <<'PY' > gpurun_out/region_$tag.txt
import csv, sys, glob
tag = sys.argv[1]
res = {}
for lanes in (2, 1):
    f = glob.glob(f"gpurun_out/rk_{tag}_{lanes}/**/rk_kernel_stats.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void sh::", "").replace("sh::", "")
        if any(t in n for t in ("k_conv3_ldr16", "_pp<", "k_upconv")):
            res.setdefault(n, {})[lanes] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
print("%-40s %12s %12s %8s" % ("kernel", "two lanes us", "one lane us", "ratio"))
tot = {1: 0.0, 2: 0.0}
per_pass = {"k_conv3_ldr16<0, 0, 0>": 9, "k_conv3_ldr16<0, 4, 0>": 2, "k_conv3_ldr16<0, 0, 1>": 2}
for n, d in sorted(res.items()):
    if 1 in d and 2 in d:
        print("%-40s %12.1f %12.1f %8.2f" % (n, d[2][0], d[1][0], d[2][0] / d[1][0]))
        for l in (1, 2): tot[l] += d[l][0] * per_pass.get(n, 1)
print("sum over a pass: two lanes %.1f us, one lane %.1f us" % (tot[2], tot[1]))
PY
rm -rf gpurun_out/rk_${tag}_1 gpurun_out/rk_${tag}_2
cat gpurun_out/region_$tag.txt
