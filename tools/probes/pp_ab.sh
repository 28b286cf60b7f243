#!/bin/bash
# interleaved A/B of the network alone: tools/probes/pp_ab.sh <rounds> <arm> ...   arm = "base", "VAR=val" (environment) or a library path
R=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for arm in "$@"; do
    tag=$(basename $arm .so)
    unset SHOULDER_LIB; E=""
    case $arm in base) ;; *.so) export SHOULDER_LIB=$PWD/$arm ;; *) E="$arm" ;; esac
    env $E python3 tools/bench_unet.py --unet ${UNET:-bf16} --reps ${REPS:-5} --layers > gpurun_out/ab/pp_${tag}_r$r.log 2>&1 || { echo "FAILED $arm"; tail -5 gpurun_out/ab/pp_${tag}_r$r.log; exit 1; }
    echo "$tag r$r $(grep 512,512 gpurun_out/ab/pp_${tag}_r$r.log | sed 's/.*device_ms_per_forward/ms/' | cut -c1-12) $(grep layers_ms gpurun_out/ab/pp_${tag}_r$r.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['layers_ms']; print({k.replace('unet.',''):v for k,v in d.items() if k in ('unet.enc0b','unet.dec0a','unet.dec0b')})")"
  done
done
