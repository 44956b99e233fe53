#!/bin/bash
# round-3 pass bb: a walk hands over to TEST as soon as it holds tune[6] - 1 candidates (the nearest hit so far then prunes the rest of the walk): 0 (=1 candidate) .. 6 (list nearly full, the product)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bb
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "cover or tiny or moving or mixed" > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2; do
for e in 1 2 3 4 5; do run 24,32,1,32,28,1,$e,1,16 $rep "" 20 cover; done
for e in 1 2 3 4; do run 24,32,1,32,28,1,$e,1,20 $rep "--config 4" 8 c4; done
for e in 1 2 3 4; do run 16,48,1,1,1,1,$e,1,32 $rep "--scene mesh" 8 mesh; done
done
