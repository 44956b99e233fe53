#!/bin/bash
# round-3 pass be: stage statistics of the build with the hand-over at 3 candidates (single launch, cover)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03be
rm -rf $OUT; mkdir -p $OUT
export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so
for t in 24,32,1,32,28,1,3,1,16 24,32,1,32,28,1,7,1,16; do
  timeout 400 python bench.py --steps 2 --warmup 1 --chain 1 --no-extras --no-cpu-baseline --tune $t > $OUT/bench_$t.json 2> $OUT/stats_$t.txt
  echo "== $t"; grep "\[stats\]" $OUT/stats_$t.txt | tail -45 | head -21
done
