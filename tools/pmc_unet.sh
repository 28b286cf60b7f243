#!/bin/bash
# SQ counters of the network alone (tools/bench_unet.py) for one library build: tools/pmc_unet.sh <tag> [<lib>]
# -> gpurun_out/pmc_<tag>.json (tools/pmc_sq.py summary)
set -e
TAG=$1; LIB=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "$LIB" ]; then export SHOULDER_LIB=$PWD/$LIB; fi
O=gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O -o sq -- python3 tools/bench_unet.py --unet ${UNET:-bf16} --reps 2 > $O/run.log 2>&1
python3 tools/pmc_sq.py $(find $O -name "*counter_collection.csv" | head -1) 64 ${UNET:-bf16} > gpurun_out/pmc_$TAG.json
python3 - <<PY
import json
d=json.load(open("gpurun_out/pmc_$TAG.json"))["kernels"]
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get("kernel_cycles",0)*kv[1]["launches"])[:14]:
    print(k[:60].ljust(60), v["launches"], "cyc", int(v.get("kernel_cycles",0)), "busy", v.get("mfma_busy_frac"), "park", v.get("wave_frac_parked"), "stall", v.get("wave_frac_issue_stalled"), "issue", v.get("wave_frac_issuing"))
PY
