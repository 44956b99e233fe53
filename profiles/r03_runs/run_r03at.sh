#!/bin/bash
# round-3 pass at: timing experiment - slices whose consecutive tickets are pixels far apart (every wave gets its share of the expensive pixels) against tiles
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03at
rm -rf $OUT; mkdir -p $OUT
for lib in tiles scatter; do
  if [ $lib = tiles ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$lib.so; fi
  for c in 2 3; do
  timeout 900 python profiles/emulate_tile_split.py --config $c --slices 1,2,4,8 > $OUT/tiles_c${c}_$lib.json 2> $OUT/err_c${c}_$lib.log
  python -c "
import json; d=json.load(open('$OUT/tiles_c${c}_$lib.json')); print('$lib c$c', {k:(v['slowest_ms'], [round(x,1) for x in v['kernel_ms_per_slice']]) for k,v in d['slices'].items()})"
  done
done
