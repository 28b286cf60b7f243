#!/bin/bash
# the CU reserve of the persistent UNet grids (SHOULDER_CU_RESERVE, read once per process), interleaved on one box:
#   tools/probes/reserve_ab.sh <rounds> <reserve> ...      STEPS (default 100) steps per timed region
R=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for cu in "$@"; do
    SHOULDER_CU_RESERVE=$cu timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-100} --warmup 4 > gpurun_out/ab/res_${cu}_r$r.log 2>&1 || { echo "FAILED $cu"; tail -5 gpurun_out/ab/res_${cu}_r$r.log; exit 1; }
    echo "reserve $cu r$r $(tail -1 gpurun_out/ab/res_${cu}_r$r.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
