#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats, the two HBM counter passes and the SQ (MFMA busy) pass of the default
# bench command.  Outputs under gpurun_out/prof_$RND/ (RND defaults to r03); summarise afterwards with tools/rocpd_stats.py / tools/pmc_traffic.py /
# tools/pmc_sq.py and copy the summaries into profiles/.
# Counters are collected in their own runs (no trace domains beside --pmc), the program directly after `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RND=${RND:-r05}
O=gpurun_out/prof_$RND
UNET=${1:-bf16}
mkdir -p $O
python3 bench.py --unet $UNET > $O/bench_$UNET.log 2>&1
tail -1 $O/bench_$UNET.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$UNET -o stats -- python3 bench.py --unet $UNET --steps 20 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof_$UNET.log 2>&1
echo stats done
# the same with ONE lane: kernel durations without another stream's kernels on the CUs (the geometry table of bench.py is a one-lane pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1_$UNET -o stats -- python3 bench.py --unet $UNET --steps 20 --warmup 2 --no-cpu-baseline --no-extra-legs --lanes 1 > $O/bench_under_rocprof_lanes1_$UNET.log 2>&1
echo stats one lane done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$UNET -o fetch -- python3 bench.py --unet $UNET --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/fetch_$UNET.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$UNET -o write -- python3 bench.py --unet $UNET --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/write_$UNET.log 2>&1
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $O/sq_$UNET -o sq -- python3 bench.py --unet $UNET --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --lanes 1 > $O/sq_$UNET.log 2>&1
echo sq done
find $O -name "*.csv" | head -30
# BASELINE configs[4]: the network alone, fp16 MFMA conv (tools/bench_unet.py --unet f16): kernel stats + HBM / SQ counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_f16_stats -o stats -- python3 tools/bench_unet.py --unet f16 --reps 5 > $O/unet_f16_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/unet_f16_fetch -o fetch -- python3 tools/bench_unet.py --unet f16 --reps 2 > $O/unet_f16_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/unet_f16_write -o write -- python3 tools/bench_unet.py --unet f16 --reps 2 > $O/unet_f16_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $O/unet_f16_sq -o sq -- python3 tools/bench_unet.py --unet f16 --reps 2 > $O/unet_f16_sq.log 2>&1
python3 tools/bench_unet.py --unet f16 --layers > $O/unet_f16.log 2>&1
python3 tools/bench_unet.py --unet bf16 --layers > $O/unet_bf16.log 2>&1
python3 tools/bench_unet.py --unet f32x --layers --reps 2 > $O/unet_f32x.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_f32x_stats -o stats -- python3 tools/bench_unet.py --unet f32x --reps 2 > $O/unet_f32x_under_rocprof.log 2>&1
echo unet f16 done
