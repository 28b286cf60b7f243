# the UNet pass of ONE lane on the two-lane grid (224 CUs, nothing beside it) against the same lane on 256 CUs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 0 1; do
export SH_BENCH_FORCE_TURNS=$f
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlg$f -o tl -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --lanes 1 > gpurun_out/tlg_bench$f.json 2>> gpurun_out/tl_err.log
python tools/unet_gaps.py gpurun_out/tlg$f/tl_kernel_trace.csv 6 > gpurun_out/tl_grid$f.txt
rm -f gpurun_out/tlg$f/*.csv
done
