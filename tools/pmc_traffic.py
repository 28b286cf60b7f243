"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv):
    python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv B unet > profiles/r01_pmc_traffic_b64_bf16.json
hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: counter unit KB; on gfx950 FETCH_SIZE reports half of the bytes of
16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section)."""
import csv, json, re, sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        n = re.sub(r"\.kd$", "", n)
        acc[n][0] += 1
        acc[n][1] += float(r["Counter_Value"])
    return acc


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`. "
                "Counter unit KB. hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half of the bytes of "
                "16-B/lane streaming reads (MI355X_MICROARCH.md, HBM). Narrower access patterns (the fp64 geometry kernels) are uncalibrated.",
       "config": {"batch": int(sys.argv[3]), "unet": sys.argv[4]}, "kernels": {}}
for k in f:
    n = f[k][0]
    fk, wk = f[k][1] / n, (w[k][1] / w[k][0] if k in w and w[k][0] else 0.0)
    out["kernels"][k] = {"launches": n, "FETCH_SIZE_KB_per_launch": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
                         "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
json.dump(out, sys.stdout, indent=1)
