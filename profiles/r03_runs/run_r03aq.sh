#!/bin/bash
# round-3 pass aq: rare shading classes (general-Standard, dielectric) held back until tune[6]/64 of the live lanes want them, for at most tune[7] trips
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03aq
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "cover or tiny or moving or slices" > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
for rep in 1 2; do
for t in 24,32,1,32,28,1,1,1,16 24,32,1,32,28,1,4,1,16 24,32,1,32,28,1,4,2,16 24,32,1,32,28,1,8,1,16 24,32,1,32,28,1,8,2,16 24,32,1,32,28,1,8,3,16 24,32,1,32,28,1,12,2,16 24,32,1,32,28,1,16,2,16 24,32,1,32,28,1,16,4,16 24,32,1,24,28,1,8,2,16; do run $t $rep "" 20 cover; done
done
