#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/collect.sh r01'
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2..: one --pmc run each (counters are never combined
# with tracing domains; FETCH_SIZE and WRITE_SIZE need separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); profiles/summarize.py turns it into the committed summary.
TAG=${1:-r01}
CHAIN=${2:-10}          # batches per launch: 10 = what the driver's `bench.py --gpus 1 --steps 20 --warmup 5` runs (two chains of 10)
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# bench.py with chains of $CHAIN batches per launch - the chain length of the driver's command, so that roofline.traffic in the driver's line is a
# measurement at that very chain length and not an extrapolation; the counter passes profile ONE chain launch after the probe
# Context flags 0, like the driver's run: the threshold measurement is on (round 3 profiled with it off).  Its probes - ten 4-sample launches of this same
# kernel, ~1.3 ms each, enqueued in front of the first batch - show up in the trace next to the one cost probe; profiles/summarize.py tells them from the batch
# launches by their duration, and the counter passes read the LAST dispatch (the chain launch).
BENCH="python $REPO/bench.py --steps $((2 * CHAIN)) --warmup 0 --chain $CHAIN --no-cpu-baseline --no-extras"
ONE="python $REPO/bench.py --steps $CHAIN --warmup 0 --chain $CHAIN --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $ONE > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $ONE > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq1 -o bench -- $ONE > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2 -o bench -- $ONE > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq3 -o bench -- $ONE > $OUT/pmc_sq3.log 2>&1
# the post passes (combine / finalize / combine_finalize / reduce_metrics (+ async fold) / add): one kernel trace per frame size, 22 launches of each kernel
for SIZE in 1920x1080 3840x2160; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/post_$SIZE -o bench -- python $REPO/bench.py --post-only $SIZE > $OUT/post_$SIZE.log 2>&1
done
grep -h '^{' $OUT/trace.log $OUT/pmc_*.log | tail -1 > $OUT/bench_line.json
find $OUT -name '*.csv' | head -40
