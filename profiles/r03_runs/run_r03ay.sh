#!/bin/bash
# round-3 pass ay: does a few-samples-per-pixel launch rank stage thresholds like the full workload does?  (feasibility of tuning them per scene with a probe)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ay
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py --scene $3 --spp $4 --chain 1 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$3_$4_$1.json 2>$OUT/err.log; python -c "
import json; d=json.load(open('$OUT/$3_$4_$1.json')); print('$3 spp $4 $1', d['value'], d['kernel_ms_per_step'])" || tail -2 $OUT/err.log; }
for scene in cover textured mixed volumes stress; do
for spp in 2 4 8 64; do
for t in 16,48,1,1,1,1,1,1,16 24,32,1,32,28,1,1,1,16 16,48,1,1,1,32,1,1,16; do
  run $t 0 $scene $spp
done; done; done
