#!/bin/bash
# round-3 pass br: post passes on full grids: their tests (bit-exact vs the oracle, 2^32-operand finalize sweep), then the profile set of the final build
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03br
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_detmath.py tests/test_gpu_comm.py -q -x > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
bash profiles/collect.sh r03ze 10 > $OUT/collect.log 2>&1; tail -1 $OUT/collect.log
