#!/bin/bash
# round-3 pass u: SKY threshold 1 / 32 / 40 / 48 / 64 on the cover scene, and 1 against 32 on the other BASELINE configs
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03u
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do for s in 1 32 40 48 64; do run 16,48,1,1,$s,1,1,1,16 $rep "" 20 cover; done; done
for rep in 1 2; do for s in 1 32; do run 16,48,1,1,$s,1,1,1,16 $rep "--config 4" 8 c4; run 16,48,1,1,$s,1,1,1,16 $rep "--config 5" 8 c5; run 16,48,1,1,$s,1,1,1,16 $rep "--config 3" 2 c3; done; done
