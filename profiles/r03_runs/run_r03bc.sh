#!/bin/bash
# round-3 pass bc: hand-over count 3 built in: neighbouring parameters re-checked (walk slice, TEST threshold), and the other scene families at 3 against 7
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bc
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2; do
for t in 24,32,1,32,28,1,3,1,16 24,32,1,32,28,1,3,1,12 24,32,1,32,28,1,3,1,20 24,32,8,32,28,1,3,1,16 24,32,16,32,28,1,3,1,16 24,40,1,32,28,1,3,1,16 24,24,1,32,28,1,3,1,16; do run $t $rep "" 20 cover; done
for t in 24,32,1,32,28,1,3,1,20 24,32,1,32,28,1,3,1,16 24,32,1,32,28,1,3,1,24; do run $t $rep "--config 4" 8 c4; done
for t in 16,48,1,1,1,1,3,1,32 16,48,1,1,1,1,3,1,24 16,48,1,1,1,1,3,1,40; do run $t $rep "--scene mesh" 8 mesh; done
for t in 16,48,1,1,1,1,3,1,16 16,48,1,1,1,1,7,1,16; do run $t $rep "--scene mixed --spp 64" 8 mixed; done
for t in 24,32,1,32,28,1,3,1,16 24,32,1,32,28,1,7,1,16; do run $t $rep "--scene textured --spp 64" 8 textured; run $t $rep "--config 5" 8 c5;  done
done
