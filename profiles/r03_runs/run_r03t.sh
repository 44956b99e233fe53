#!/bin/bash
# round-3 pass t: SKY threshold (lanes waiting in the sky + fold stage before it runs), alternating runs on one box
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03t
rm -rf $OUT; mkdir -p $OUT
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --tune $1 > $OUT/cover_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/cover_$1_$2.json')); print('$1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3 4; do for t in 16,48,1,1,1,1,1,1,16 16,48,1,1,16,1,1,1,16 16,48,1,1,24,1,1,1,16 16,48,1,1,32,1,1,1,16; do run $t $rep; done; done
