#!/bin/bash
# a library variant that differs from the in-tree build only in the flags of ONE translation unit (default: unet16_pp.hip):
#   tools/probes/pp_variant.sh <name> [-DFLAG ...]   ->  ab/lib<name>.so   (TU=<file> selects another unit)
set -e
name=$1; shift
TU=${TU:-unet16_pp}
mkdir -p ab /tmp/ppv
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-rdc "$@" -c shoulder_amd/csrc/$TU.hip -o /tmp/ppv/$name.o
objs=""
for o in shoulder_amd/lib/obj/*.o; do b=$(basename $o .o); if [ "$b" = "$TU" ]; then objs="$objs /tmp/ppv/$name.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -fno-gpu-rdc -o ab/lib$name.so $objs
echo ab/lib$name.so
