#!/bin/bash
# round-3 pass bv: general family, REGEN threshold x SKY threshold (mesh 250k, mixed primitives, Cornell with volumes)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bv
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2; do
for t in 16,48,1,1,1,1,3,1 8,48,1,1,16,1,3,1 4,48,1,1,16,1,3,1 1,48,1,1,16,1,3,1 8,48,1,1,24,1,3,1 8,48,1,1,8,1,3,1 12,48,1,1,16,1,3,1; do
  run $t,24 $rep "--scene mesh" 6 mesh; run $t,16 $rep "--scene mixed --spp 64" 8 mixed; run $t,16 $rep "--scene volumes --spp 32" 6 volumes
done; done
