#!/bin/bash
# round-3 pass am: HIT threshold 16..32 x walk threshold 40 / 48 on the tiles build (cover, three alternating passes), then the best two on the other configs
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03am
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do
for t in 16,48,1,1,28,1,1,1,16 16,40,1,16,28,1,1,1,16 16,48,1,16,28,1,1,1,16 16,40,1,20,28,1,1,1,16 16,40,1,24,28,1,1,1,16 16,40,1,32,28,1,1,1,16 16,48,1,24,28,1,1,1,16 16,40,8,16,28,1,1,1,16; do run $t $rep "" 20 cover; done
done
for rep in 1 2; do for t in 16,48,1,1,28,1,1,1,16 16,40,1,16,28,1,1,1,16 16,40,1,24,28,1,1,1,16; do run $t $rep "--config 4" 8 c4; run $t $rep "--config 5" 8 c5; run $t $rep "--config 3" 2 c3; done; done
for rep in 1 2; do for t in 16,48,1,1,1,1,1,1,32 16,40,1,16,1,1,1,1,32 16,48,1,16,1,1,1,1,32; do run $t $rep "--scene mesh" 8 mesh; done; done
