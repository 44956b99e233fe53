#!/bin/bash
# round-3 pass h: walk slice (node visits per trip) and TEST / HIT thresholds against tree depth: mesh (depth 21, 26 visits per ray), stress (10 000 spheres), moving
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03h
rm -rf $OUT; mkdir -p $OUT
run() { timeout 300 python bench.py --scene $1 --steps $3 --warmup 1 --no-extras --no-cpu-baseline --tune $2 > $OUT/$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$1_$2.json')); print('$1 $2', d['value'], d['kernel_ms_per_step'], d['config']['bvh_depth'])"; }
for S in 32 48 64; do for TH in 1,1 24,8 32,8; do run mesh 16,48,$TH,1,1,1,1,$S 4; done; done
for S in 16 24 32; do for TH in 1,1 16,1; do run stress 16,48,$TH,1,1,1,1,$S 8; done; done
for S in 16 24; do run moving 16,48,1,1,1,1,1,1,$S 8; done
for S in 12 16 20; do run cover 16,48,1,1,1,1,1,1,$S 8; done
