#!/bin/bash
# round-3 pass aj: the whole GPU suite on the build with tile-numbered tickets, wide-code volume kernels and the nine-at-once finalize
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03aj
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_suite.log 2>&1; tail -5 $OUT/gpu_suite.log
