#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_lds; mkdir -p $O
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/a -o a -- python3 tools/bench_unet.py --unet bf16 --reps 2 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/b -o b -- python3 tools/bench_unet.py --unet bf16 --reps 2 > $O/b.log 2>&1
ls $O/a $O/b
