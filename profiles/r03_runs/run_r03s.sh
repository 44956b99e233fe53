#!/bin/bash
# round-3 pass s: stage thresholds on the final build, cover scene (REGEN / TRAV / SKY thresholds; the round-2 optimum was 16,48,1,1,1)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03s
rm -rf $OUT; mkdir -p $OUT
run() { timeout 300 python bench.py --steps 16 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/cover_$1.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/cover_$1.json')); print('$1', d['value'], d['kernel_ms_per_step'])"; }
for t in 16,48,1,1,1,1,1,1,16 8,48,1,1,1,1,1,1,16 24,48,1,1,1,1,1,1,16 32,48,1,1,1,1,1,1,16 16,40,1,1,1,1,1,1,16 16,56,1,1,1,1,1,1,16 16,48,1,1,8,1,1,1,16 16,48,1,1,16,1,1,1,16 16,48,4,1,1,1,1,1,16 16,48,1,4,1,1,1,1,16 24,56,1,1,8,1,1,1,16 16,48,1,1,1,1,1,1,16; do run $t; done
