cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ti2 -o tl -- python3 bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/ti_bench2.json 2>> gpurun_out/tl_err.log
export SH_BENCH_FORCE_TURNS=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ti1 -o tl -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --lanes 1 > gpurun_out/ti_bench1.json 2>> gpurun_out/tl_err.log
python tools/interference.py gpurun_out/ti2/tl_kernel_trace.csv gpurun_out/ti1/tl_kernel_trace.csv 6 > gpurun_out/interference.txt 2>&1
rm -f gpurun_out/ti1/*.csv gpurun_out/ti2/*.csv
