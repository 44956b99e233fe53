#!/bin/bash
# round-3 pass bx: at how many samples per pixel does a probe rank the three threshold families like the full workload (10 000 spheres, cover, mesh, volumes)?
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bx
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --spp $4 --chain 1 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$4_$1.json 2>$OUT/err.log; python -c "
import json; d=json.load(open('$OUT/$5_$4_$1.json')); print('$5 spp $4 $1', d['value'], d['kernel_ms_per_step'])" || tail -2 $OUT/err.log; }
for spp in 4 8 16 32 256; do
for t in 24,32,1,32,28,1,3,1 16,48,1,1,1,1,3,1 8,48,1,1,8,1,3,1; do
  run $t,16 0 "--config 4" $spp c4; run $t,16 0 "" $spp cover
done; done
for spp in 4 16 32; do
for t in 24,32,1,32,28,1,3,1 16,48,1,1,1,1,3,1 8,48,1,1,8,1,3,1; do
  run $t,24 0 "--scene mesh" $spp mesh; run $t,16 0 "--scene volumes" $spp volumes
done; done
