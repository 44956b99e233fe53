#!/bin/bash
# round-3 pass ah: ticket chunks as 8 x 8 tiles (product) against 64 x 1 strips (A/B library), alternating runs on one box; parity subset on the tiles build
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ah
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_golden.py -q -x > $OUT/parity.log 2>&1; tail -3 $OUT/parity.log
run() { # lib tag, rep, bench args, steps, name
  if [ $1 = tiles ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$1.so; fi
  timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do for lib in tiles strips; do run $lib $rep "" 20 cover; done; done
for rep in 1 2; do for lib in tiles strips; do run $lib $rep "--config 4" 8 c4; run $lib $rep "--config 5" 8 c5; run $lib $rep "--config 3" 2 c3; run $lib $rep "--scene mesh" 8 mesh; done; done
