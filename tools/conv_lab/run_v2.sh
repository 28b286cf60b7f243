#!/bin/bash
L=tools/conv_lab/conv_lab
K=${1:-v2}
timeout -k 5 60 $L 64 64 64 64 2 2 $K 1 && timeout -k 5 60 $L 256 256 64 64 64 10 $K 1 && timeout -k 5 60 $L 128 128 128 128 64 10 $K 1 && timeout -k 5 60 $L 64 64 256 256 64 10 $K 1 && timeout -k 5 60 $L 32 32 512 512 64 10 $K 1 && timeout -k 5 60 $L 256 256 128 64 64 10 $K 1
