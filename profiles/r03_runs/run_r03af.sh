#!/bin/bash
# round-3 pass af: where the finalize kernel's time goes - timing-only builds without the table fix-up (1), without the transcendentals (2), without both (3)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03af
rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in base; do
    if [ $v = base ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$v.so; fi
    for size in 1920x1080 3840x2160; do timeout 300 python bench.py --post-only $size > $OUT/post_${v}_${size}_$rep.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/post_${v}_${size}_$rep.json'))['post_passes']['$size']; print('$v $size', {k: v['GBps'] for k, v in d.items()})"; done
  done
done
