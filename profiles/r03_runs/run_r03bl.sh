#!/bin/bash
# round-3 pass bl: slices (one after the other on one GPU) under latency-oriented thresholds: everything at once, and intermediate settings
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bl
rm -rf $OUT; mkdir -p $OUT
for t in 24,32,1,32,28,1,3,1,16 1,1,1,1,1,1,3,1,16 1,1,1,1,1,1,7,1,16 8,16,1,8,8,1,3,1,16 1,1,1,1,1,1,3,1,8 1,1,1,1,1,1,3,1,32; do
  timeout 600 python profiles/emulate_tile_split.py --config 2 --slices 1,4,8 --tune $t > $OUT/tiles_c2_$t.json 2> $OUT/err.log
  python -c "
import json; d=json.load(open('$OUT/tiles_c2_$t.json')); print('$t', {k:(v['slowest_ms'], [round(x,1) for x in v['kernel_ms_per_slice']]) for k,v in d['slices'].items()})"
done
