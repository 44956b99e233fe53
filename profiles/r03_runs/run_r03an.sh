#!/bin/bash
# round-3 pass an: HIT threshold 32..56 x walk threshold 32..48, TEST threshold 8 / 16 / 24 on top (cover, three alternating passes); best on the other configs
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03an
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do
for t in 16,40,1,32,28,1,1,1,16 16,40,1,40,28,1,1,1,16 16,40,1,48,28,1,1,1,16 16,40,1,56,28,1,1,1,16 16,32,1,32,28,1,1,1,16 16,48,1,32,28,1,1,1,16 16,32,1,40,28,1,1,1,16 16,40,8,32,28,1,1,1,16 16,40,16,32,28,1,1,1,16 16,40,24,32,28,1,1,1,16 16,40,1,32,20,1,1,1,16 16,40,1,32,36,1,1,1,16 24,40,1,32,28,1,1,1,16 8,40,1,32,28,1,1,1,16; do run $t $rep "" 20 cover; done
done
for rep in 1 2; do for t in 16,40,1,32,28,1,1,1,16 16,40,1,40,28,1,1,1,16; do run $t $rep "--config 4" 8 c4; run $t $rep "--config 5" 8 c5; run $t $rep "--config 3" 2 c3; done; done
