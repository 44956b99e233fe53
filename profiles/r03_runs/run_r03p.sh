#!/bin/bash
# round-3 pass p: what a triangle-only scene kind could gain on the 250 882-triangle mesh (timing-only build -DRTOW_EXPERIMENT_TRIANGLES_ONLY: general_hit without type dispatch / transform code)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03p
rm -rf $OUT; mkdir -p $OUT
T=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_tri.so
ARGS="--scene mesh --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
for rep in 1 2; do for fl in 0 2; do
  timeout 300 python bench.py $ARGS --context-flags $fl > $OUT/product_f${fl}_$rep.json 2>> $OUT/err
  RTOW_LIB_PATH=$T timeout 300 python bench.py $ARGS --context-flags $fl > $OUT/tri_f${fl}_$rep.json 2>> $OUT/err
done; done
for f in $OUT/*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'])"; done
