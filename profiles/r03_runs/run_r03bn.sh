#!/bin/bash
# round-3 pass bn: wave priority per stage re-checked on the final build: exact tests at 1 (product, 16) against none (0), + HIT (80), + REGEN (17), + SKY (272)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bn
rm -rf $OUT; mkdir -p $OUT
run() { if [ $1 = prio16 ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$1.so; fi
  timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do for lib in prio16 prio0 prio80 prio17 prio272; do run $lib $rep "" 20 cover; done; done
for rep in 1 2; do for lib in prio16 prio0 prio80; do run $lib $rep "--config 4" 8 c4; run $lib $rep "--scene mesh" 8 mesh; done; done
