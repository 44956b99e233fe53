#!/bin/bash
# round-3 pass ao: confirmation of the new thresholds (HIT from half of the live lanes, walk from 1/2 .. 5/8) on every BASELINE config and the mesh
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ao
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do for t in 16,48,1,1,28,1,1,1,16 16,40,1,32,28,1,1,1,16 16,32,1,32,28,1,1,1,16 24,32,1,32,28,1,1,1,16 16,36,1,28,28,1,1,1,16; do run $t $rep "" 20 cover; run $t $rep "--config 4" 8 c4; done; done
for rep in 1 2; do for t in 16,48,1,1,28,1,1,1,16 16,40,1,32,28,1,1,1,16 16,32,1,32,28,1,1,1,16 24,32,1,32,28,1,1,1,16; do run $t $rep "--config 5" 8 c5; run $t $rep "--config 3" 2 c3; done; done
for rep in 1 2; do for t in 16,48,1,1,1,1,1,1,32 16,48,1,32,1,1,1,1,32 16,40,1,32,1,1,1,1,32; do run $t $rep "--scene mesh" 8 mesh; done; done
