#!/bin/bash
# round-3 pass au: soak of the tiles build - chained launches against sequences (GPU against GPU, every pixel) and whole frames against the oracle
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03au
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python tests/soak_chain.py 1.0 > $OUT/soak_chain.log 2>&1; tail -25 $OUT/soak_chain.log
timeout 1500 python tests/soak_frames.py 1.0 > $OUT/soak_frames.log 2>&1; tail -32 $OUT/soak_frames.log
