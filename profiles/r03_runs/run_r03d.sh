#!/bin/bash
# round-3 pass d: what regrouping rays between lanes could win - measured, not estimated.
#   coherent_waves  all 64 lanes of a chunk trace the chunk's first pixel (wrong image): every stage runs with 64 lanes -> ceiling of ANY regrouping
#   double_test     the exact sphere test evaluated twice (same image): product time - this = wall-time cost of the TEST arithmetic
#   double_walk     the slab arithmetic of a node visit evaluated twice (same image): likewise for the walk
#   stats           wave-level stage statistics of the product kernel (lane populations per stage, wave time per stage)
# Builds:  make OBJDIR=build/ab_<x> LIB=build/librtow_hip_<x>.so EXTRA=-DRTOW_EXPERIMENT_<X>   and   make stats
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03d
rm -rf $OUT; mkdir -p $OUT
B=$REPO/raytracing-in-one-weekend_amd/csrc/build
ARGS="--steps 16 --warmup 1 --no-cpu-baseline --no-extras"
for rep in 1 2 3; do
  timeout 200 python bench.py $ARGS > $OUT/bench_product_$rep.json 2>> $OUT/bench.err
  for x in coherent_waves double_test double_walk; do
    RTOW_LIB_PATH=$B/librtow_hip_$x.so timeout 300 python bench.py $ARGS > $OUT/bench_${x}_$rep.json 2>> $OUT/bench.err
  done
done
# single launches too (no chain): the stats build reports per launch
RTOW_LIB_PATH=$B/librtow_hip_stats.so timeout 300 python bench.py --steps 1 --warmup 1 --chain 1 --no-cpu-baseline --no-extras > $OUT/bench_stats.json 2> $OUT/stats.err
for f in $OUT/bench_*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'], d['rays_per_sample'], d['mrays_per_s'])"; done
grep "\[stats\]" $OUT/stats.err | grep -v "wave #" | tail -40
