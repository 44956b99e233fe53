#!/bin/bash
# round-3 pass bf: thresholds around the defaults once more with the hand-over at 3 candidates (cover, three passes), goldens on the GPU
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bf
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_golden.py -q -x > $OUT/golden.log 2>&1; tail -2 $OUT/golden.log
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do
for t in 24,32,1,32,28,1,3,1,16 24,32,1,24,28,1,3,1,16 24,32,1,40,28,1,3,1,16 16,32,1,32,28,1,3,1,16 32,32,1,32,28,1,3,1,16 24,32,1,32,20,1,3,1,16 24,32,1,32,36,1,3,1,16 24,24,1,32,28,1,3,1,16 24,24,1,40,28,1,3,1,16; do run $t $rep "" 20 cover; done
done
