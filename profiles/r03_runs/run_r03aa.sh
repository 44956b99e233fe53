#!/bin/bash
# round-3 pass aa: coordinate sweep of the stage thresholds around the new default (16,48,1,1,32), cover scene, alternating repeats
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03aa
rm -rf $OUT; mkdir -p $OUT
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --tune $1 > $OUT/cover_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/cover_$1_$2.json')); print('$1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do for t in 16,48,1,1,32,1,1,1,16 12,48,1,1,32,1,1,1,16 20,48,1,1,32,1,1,1,16 16,44,1,1,32,1,1,1,16 16,52,1,1,32,1,1,1,16 16,48,2,1,32,1,1,1,16 16,48,1,2,32,1,1,1,16 16,48,1,1,28,1,1,1,16 16,48,1,1,36,1,1,1,16 16,48,1,1,32,1,1,1,14 16,48,1,1,32,1,1,1,18; do run $t $rep; done; done
