#!/bin/bash
# round-3 pass ai: tile shape of the 64-ticket chunks: 8 x 8 (product) against 4 x 16, 16 x 4, 32 x 2 (A/B libraries), alternating runs on one box
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ai
rm -rf $OUT; mkdir -p $OUT
run() { # lib tag, rep, bench args, steps, name
  if [ $1 = tw8 ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$1.so; fi
  timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do for lib in tw8 tw4 tw16 tw32; do run $lib $rep "" 20 cover; done; done
for rep in 1 2; do for lib in tw8 tw4 tw16; do run $lib $rep "--config 4" 8 c4; run $lib $rep "--config 5" 8 c5; run $lib $rep "--scene mesh" 8 mesh; done; done
