#!/bin/bash
# round-3 pass ak: scheduler thresholds re-swept on the build with tile-numbered tickets (cover scene, driver's chain; one change at a time around 16,48,1,1,28,1,1,1,16)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ak
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py --steps 20 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/cover_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/cover_$1_$2.json')); print('$1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2; do
for t in 16,48,1,1,28,1,1,1,16 8,48,1,1,28,1,1,1,16 24,48,1,1,28,1,1,1,16 32,48,1,1,28,1,1,1,16 16,40,1,1,28,1,1,1,16 16,56,1,1,28,1,1,1,16 16,48,1,1,20,1,1,1,16 16,48,1,1,36,1,1,1,16 16,48,8,1,28,1,1,1,16 16,48,1,8,28,1,1,1,16 16,48,1,1,28,1,1,1,12 16,48,1,1,28,1,1,1,14 16,48,1,1,28,1,1,1,18 16,48,1,1,28,1,1,1,20 16,48,1,1,28,1,1,1,24; do run $t $rep; done
done
