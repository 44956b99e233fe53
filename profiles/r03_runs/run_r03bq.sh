#!/bin/bash
# round-3 pass bq: grid size of the post passes: combine / finalize at 1024 / 2048 / 4096 / 8192 blocks / one block per 256 pixels; add at 16384 / 65536 / one per 256 elements
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bq
rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in base p1024 p4096 p8192 pfull; do
    if [ $v = base ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$v.so; fi
    for size in 1920x1080 3840x2160; do timeout 300 python bench.py --post-only $size > $OUT/post_${v}_${size}_$rep.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/post_${v}_${size}_$rep.json'))['post_passes']['$size']; print('$v $size', {k: v['GBps'] for k, v in d.items()})"; done
  done
done
