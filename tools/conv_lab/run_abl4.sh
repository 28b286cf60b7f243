#!/bin/bash
L=tools/conv_lab/conv_lab
for K in base v4_100 v4a1 v4a2 v4a3; do
  timeout -k 5 60 $L 256 256 64 64 64 10 $K 0 || exit 1
  timeout -k 5 60 $L 64 64 256 256 64 10 $K 0 || exit 1
done
echo zero-data
LAB_ZERO=1 timeout -k 5 60 $L 64 64 256 256 64 10 v4_100 0
LAB_ZERO=1 timeout -k 5 60 $L 64 64 256 256 64 10 v4a2 0
