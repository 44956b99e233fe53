#!/bin/bash
# round-3 pass bu: the general family's thresholds re-swept with the hand-over at 3 candidates (mesh 250k, mixed primitives)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bu
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2; do
for t in 16,48,1,1,1,1,3,1 16,40,1,1,1,1,3,1 16,32,1,1,1,1,3,1 24,48,1,1,1,1,3,1 8,48,1,1,1,1,3,1 16,48,1,8,1,1,3,1 16,48,1,1,16,1,3,1 16,48,1,1,1,1,4,1 16,56,1,1,1,1,3,1; do
  run $t,24 $rep "--scene mesh" 6 mesh; run $t,16 $rep "--scene mixed --spp 64" 8 mixed
done; done
