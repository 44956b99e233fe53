#!/bin/bash
# round-3 pass ax: stage thresholds for the general-entity kinds on scenes that live in LDS (mixed, volumes, textured) and the fog mesh: their own (16,48,1,1,1) against the sphere kinds'
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ax
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py --scene $3 --spp $6 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>$OUT/err.log; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])" || tail -2 $OUT/err.log; }
for rep in 1 2; do
for t in 16,48,1,1,1,1,1,1,16 24,32,1,32,28,1,1,1,16 16,48,1,32,1,1,1,1,16 16,32,1,16,1,1,1,1,16 24,32,1,32,28,32,1,1,16; do
  run $t $rep mixed 8 mixed 64; run $t $rep textured 8 textured 64; run $t $rep volumes 8 volumes 32
done
for t in 16,48,1,1,1,1,1,1,32 16,48,1,1,1,32,1,1,32; do run $t $rep meshfog 4 meshfog 16; done
done
