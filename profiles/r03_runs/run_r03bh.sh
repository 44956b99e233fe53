#!/bin/bash
# round-3 pass bh: soak of the final build (hand-over at 3 candidates exercises the walk's resume-with-best path on every ray): whole frames, chains, fuzz
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bh
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python tests/soak_frames.py 1.0 > $OUT/soak_frames.log 2>&1; tail -2 $OUT/soak_frames.log
timeout 1200 python tests/soak_chain.py 1.0 > $OUT/soak_chain.log 2>&1; tail -1 $OUT/soak_chain.log
RTOW_FUZZ_SEEDS=6000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log
RTOW_FUZZ_SEEDS=2000 RTOW_FUZZ_HEAVY=1 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x > $OUT/fuzz_heavy.log 2>&1; tail -2 $OUT/fuzz_heavy.log
