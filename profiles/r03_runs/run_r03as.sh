#!/bin/bash
# round-3 pass as: predicted tile split (slices one after the other on one GPU) under the old and the new stage thresholds: does waiting in sparse waves slow the slowest slice?
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03as
rm -rf $OUT; mkdir -p $OUT
for t in 16,48,1,1,28,1,1,1,16 24,32,1,32,28,1,1,1,16; do
  timeout 600 python profiles/emulate_tile_split.py --config 2 --slices 1,2,4,8 --tune $t > $OUT/tiles_c2_$t.json 2> $OUT/err_c2_$t.log
  python -c "
import json; d=json.load(open('$OUT/tiles_c2_$t.json')); print('$t', json.dumps(d.get('summary', d['slices']))[:600])"
done
