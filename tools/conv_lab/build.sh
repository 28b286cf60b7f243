#!/bin/bash
# build the lab binary; -save-temps output goes to /tmp/conv_lab_tmp
set -e
cd "$(dirname "$0")/../.."
mkdir -p /tmp/conv_lab_tmp
( cd /tmp/conv_lab_tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I /root/repo/shoulder_amd/csrc -I /root/repo/include -I /root/repo/tools/conv_lab -save-temps -Wno-unused-value -o /root/repo/tools/conv_lab/conv_lab /root/repo/tools/conv_lab/conv_lab.hip 2>&1 | grep -E "error|warning: v" || true )
python3 /root/repo/tools/conv_lab/kinfo.py | grep conv3
