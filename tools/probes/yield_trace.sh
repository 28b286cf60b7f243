cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for y in 0 1; do
export SHOULDER_CU_YIELD=$y
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tly$y -o tl -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs > gpurun_out/tly_bench$y.json 2>> gpurun_out/tl_err.log
python tools/unet_gaps.py gpurun_out/tly$y/tl_kernel_trace.csv 6 > gpurun_out/tl_yield$y.txt
python tools/lane_timeline.py gpurun_out/tly$y/tl_kernel_trace.csv 6 >> gpurun_out/tl_yield$y.txt
rm -f gpurun_out/tly$y/*.csv
done
