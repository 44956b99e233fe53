#!/bin/bash
# round-2 pass n: ceiling of any regrouping of the rare shading classes - a build in which every surface shades as lambert (wrong image, timing only)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02n
rm -rf $OUT; mkdir -p $OUT
B=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_all_lambert.so
for rep in 1 2 3; do
  timeout 200 python bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_product_$rep.json 2>> $OUT/bench.err
  RTOW_LIB_PATH=$B timeout 200 python bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_all_lambert_$rep.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'], d['rays_per_sample'], d['mrays_per_s'])"; done
