#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats + the two HBM counter passes of the default bench command.
# Outputs under gpurun_out/prof_r01/; summarise afterwards with tools/rocpd_stats.py and tools/pmc_traffic.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r01
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
echo write done
python3 bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-400
find $O -name "*.csv" | head -20
