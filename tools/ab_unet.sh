#!/bin/bash
# A/B of library builds on one GPU box: tools/ab_unet.sh <rounds> <lib> [<lib> ...]  ("base" = the in-tree library).
# Every round runs tools/bench_unet.py once per library, in the order given (interleaved rounds, same device).
R=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    if [ "$lib" = base ]; then unset SHOULDER_LIB; else export SHOULDER_LIB=$PWD/$lib; fi
    python3 tools/bench_unet.py --unet ${UNET:-bf16} --reps ${REPS:-5} --layers > gpurun_out/ab/${tag}_r$r.log 2>&1 || { echo "FAILED $tag"; tail -5 gpurun_out/ab/${tag}_r$r.log; exit 1; }
    echo "$tag r$r $(grep 512,512 gpurun_out/ab/${tag}_r$r.log | sed 's/.*device_ms_per_forward/ms/' )"
  done
done
