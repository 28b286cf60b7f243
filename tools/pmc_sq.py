"""Per-kernel SQ counters from one rocprofv3 --pmc pass (--output-format csv) of bench.py:
    python tools/pmc_sq.py sq_counter_collection.csv B unet > profiles/r02_pmc_sq_b64_bf16.json
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): the counter sums, over every SIMD of
the chip, the cycles its matrix pipe is busy; GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS
note), so GRBM_GUI_ACTIVE / 8 is the kernel's duration in shader cycles.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY
count quad-cycles summed over waves: their ratios say where a wave's lifetime goes (parked on s_waitcnt / barrier, issue
stalled, issuing)."""
import csv, json, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
    n = re.sub(r"\.kd$", "", n)
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[n][r["Counter_Name"]] += 1
out = {"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY "
                "SQ_INSTS_VALU_MFMA_MOPS_* GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --lanes 1 "
                "(one lane: the counters of a dispatch are not mixed with another stream's kernels). Values are per launch. "
                "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs).",
       "config": {"batch": int(sys.argv[2]), "unet": sys.argv[3]}, "kernels": {}}
for k, c in acc.items():
    launches = max(cnt[k].values())
    per = {name: v / cnt[k][name] for name, v in c.items()}
    e = {"launches": launches}
    e.update({name: round(v, 1) for name, v in per.items()})
    cyc = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in per:
        e["kernel_cycles"] = round(cyc, 1)
        e["mfma_busy_frac"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        for name, key in (("SQ_WAIT_ANY", "wave_frac_parked"), ("SQ_WAIT_INST_ANY", "wave_frac_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "wave_frac_issuing")):
            if name in per:
                e[key] = round(per[name] / wc, 4)
    out["kernels"][k] = e
json.dump(out, sys.stdout, indent=1)
