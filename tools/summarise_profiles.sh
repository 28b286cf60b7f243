#!/bin/bash
# After tools/collect_profiles.sh (its outputs merged back under gpurun_out/prof_$RND/): write the summaries the judge reads into profiles/.
set -e
RND=${RND:-r05}
P=gpurun_out/prof_$RND
cp $P/stats_bf16/stats_kernel_stats.csv profiles/${RND}_kernel_stats_b64_bf16.csv
cp $P/stats1_bf16/stats_kernel_stats.csv profiles/${RND}_kernel_stats_b64_bf16_one_lane.csv
cp $P/unet_f32x_stats/stats_kernel_stats.csv profiles/${RND}_kernel_stats_unet_alone_f32x.csv
cp $P/unet_f16_stats/stats_kernel_stats.csv profiles/${RND}_kernel_stats_unet_alone_f16.csv
python tools/pmc_traffic.py $P/fetch_bf16/fetch_counter_collection.csv $P/write_bf16/write_counter_collection.csv 64 bf16 > profiles/${RND}_pmc_traffic_b64_bf16.json
python tools/pmc_sq.py $P/sq_bf16/sq_counter_collection.csv 64 bf16 > profiles/${RND}_pmc_sq_b64_bf16.json
python tools/pmc_traffic.py $P/unet_f16_fetch/fetch_counter_collection.csv $P/unet_f16_write/write_counter_collection.csv 64 f16 > profiles/${RND}_pmc_traffic_unet_alone_f16.json
python tools/pmc_sq.py $P/unet_f16_sq/sq_counter_collection.csv 64 f16 > profiles/${RND}_pmc_sq_unet_alone_f16.json
grep "^{\"metric" $P/bench_bf16.log | tail -1 > profiles/${RND}_bench_b64_bf16.json
grep "^{\"metric" $P/bench_under_rocprof_bf16.log | tail -1 > profiles/${RND}_bench_b64_bf16_under_rocprof.json
(grep '^{' $P/unet_bf16.log; grep '^{' $P/unet_f16.log; grep '^{' $P/unet_f32x.log) > profiles/${RND}_bench_unet_alone.jsonl
