#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/collect.sh r05 10'                                   the driver's command (C2 cover scene, chains of 10) + the post passes
#   gpurun -- 'POST=0 bash profiles/collect.sh r05_c4 10 --config 4'              any other bench.py workload: everything after the chain length goes to bench.py
#   gpurun -- 'POST=0 bash profiles/collect.sh r05_mesh 4 --scene mesh'           (--config N, --scene S, --depth D, --tune ..., --context-flags F, --diag-stride 16 ...)
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2..: one --pmc run each (counters are never combined
# with tracing domains; FETCH_SIZE and WRITE_SIZE need separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# A pass whose counters this box does not know fails by itself and is skipped by profiles/summarize.py.
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); profiles/summarize.py turns it into the committed summary.
TAG=${1:-r01}
CHAIN=${2:-10}          # batches per launch: 10 = what the driver's `bench.py --gpus 1 --steps 20 --warmup 5` runs (two chains of 10)
shift; shift
EXTRA="$*"
POST=${POST:-1}         # 0: skip the post-pass traces
L2=${L2:-0}             # 1: add the L2 / texture-cache passes (scenes whose tree is read from HBM)
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# bench.py with chains of $CHAIN batches per launch - the chain length of the driver's command, so that roofline.traffic in the driver's line is a
# measurement at that very chain length and not an extrapolation; the counter passes profile ONE chain launch after the probe
# Context flags 0, like the driver's run: the threshold measurement is on (round 3 profiled with it off).  Its probes - ten 4-sample launches of this same
# kernel, ~1.3 ms each, enqueued in front of the first batch - show up in the trace next to the one cost probe; profiles/summarize.py tells them from the batch
# launches by their duration, and the counter passes read the LAST dispatch (the chain launch).
BENCH="python $REPO/bench.py --steps $((2 * CHAIN)) --warmup 0 --chain $CHAIN --no-cpu-baseline --no-extras $EXTRA"
ONE="python $REPO/bench.py --steps $CHAIN --warmup 0 --chain $CHAIN --no-cpu-baseline --no-extras $EXTRA"
echo "$BENCH" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
pmc() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o bench -- $ONE > $OUT/pmc_$name.log 2>&1 || echo "pass $name failed" >> $OUT/failed_passes.txt; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
pmc sq3 SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
# what the wave cycles that wait for an instruction wait for: instruction fetch, the instruction cache, and the LDS / vector-memory / scalar queues
pmc sq4 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
pmc sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
pmc sq5 SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU SQ_WAVE32_INSTS SQ_INSTS_VALU_TRANS SQ_INSTS_EXP_GDS SQ_INSTS_WAVE32_LDS
if [ "$L2" = "1" ]; then
  pmc tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  pmc tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  pmc tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
fi
if [ "$POST" = "1" ]; then
  # the post passes (combine / finalize / combine_finalize / reduce_metrics (+ async fold) / add): one kernel trace per frame size, 22 launches of each kernel
  for SIZE in 1920x1080 3840x2160; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/post_$SIZE -o bench -- python $REPO/bench.py --post-only $SIZE > $OUT/post_$SIZE.log 2>&1
  done
fi
grep -h '^{' $OUT/trace.log $OUT/pmc_*.log | tail -1 > $OUT/bench_line.json
find $OUT -name '*.csv' | wc -l; cat $OUT/failed_passes.txt 2>/dev/null
