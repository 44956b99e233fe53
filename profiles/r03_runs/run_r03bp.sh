#!/bin/bash
# round-3 pass bp: what the accumulator add (2 reads + 1 write per float4) can reach: blocks 1024 / 2048 / 4096 / one per 256 elements, two elements per iteration, streaming hints
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bp
rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in base addv2 addv3 addb1024 addb4096 addb65536 addv2b4096; do
    if [ $v = base ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$v.so; fi
    for size in 1920x1080 3840x2160; do timeout 300 python bench.py --post-only $size > $OUT/post_${v}_${size}_$rep.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/post_${v}_${size}_$rep.json'))['post_passes']['$size']; print('$v $size add', d['add_accum']['GBps'], 'combine', d['combine']['GBps'])"; done
  done
done
