#!/bin/bash
# usage: run_many.sh kernel...   (each on the three main layer shapes, checked once on a small case)
L=tools/conv_lab/conv_lab
for K in "$@"; do
  timeout -k 5 60 $L 64 64 64 128 3 2 $K 1 | head -1 || exit 1
  timeout -k 5 60 $L 256 256 64 64 64 10 $K 0 || exit 1
  timeout -k 5 60 $L 128 128 128 128 64 10 $K 0 || exit 1
  timeout -k 5 60 $L 64 64 256 256 64 10 $K 0 || exit 1
done
