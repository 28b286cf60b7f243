#!/bin/bash
# interleaved A/B of library builds on the headline (bench.py, steady state of STEPS steps): tools/probes/lib_ab.sh <rounds> <arm> ...
#   arm = "base" (the in-tree library) or a library path; STEPS (default 100) steps per timed region
R=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for arm in "$@"; do
    tag=$(basename $arm .so)
    unset SHOULDER_LIB
    case $arm in base) ;; *) export SHOULDER_LIB=$PWD/$arm ;; esac
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-100} --warmup 4 > gpurun_out/ab/lib_${tag}_r$r.log 2>&1 || { echo "FAILED $arm"; tail -5 gpurun_out/ab/lib_${tag}_r$r.log; exit 1; }
    echo "$tag r$r $(tail -1 gpurun_out/ab/lib_${tag}_r$r.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
