#!/bin/bash
# layer shapes of the UNet at B=64: level1 64->64 @256^2, level2 128->128 @128^2, level3 256->256 @64^2
L=tools/conv_lab/conv_lab
for k in base abl0 abl1 abl2 abl3 abl4 abl5; do
  timeout -k 5 60 $L 256 256 64 64 64 10 $k $([ $k = base ] && echo 1 || echo 0) || exit 1
done
timeout -k 5 60 $L 128 128 128 128 64 10 base 0 && timeout -k 5 60 $L 64 64 256 256 64 10 base 0 && timeout -k 5 60 $L 512 512 32 32 64 10 base 0
