#!/bin/bash
# development helper: rocprofv3 passes for the bench command; outputs under gpurun_out/prof
set -x
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
CMD1="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $CMD1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $CMD1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $OUT/pmc_sq1 -o bench -- $CMD1 > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq2 -o bench -- $CMD1 > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq3 -o bench -- $CMD1 > $OUT/pmc_sq3.log 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
ls -R $OUT | head -50
