#!/bin/bash
# round-3 pass bi: stage statistics of the 250 882-triangle mesh and of the 10 000-sphere scene (single launches)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bi
rm -rf $OUT; mkdir -p $OUT
export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so
timeout 400 python bench.py --scene mesh --steps 2 --warmup 1 --chain 1 --no-extras --no-cpu-baseline > $OUT/bench_mesh.json 2> $OUT/stats_mesh.txt
echo "== mesh"; grep "\[stats\]" $OUT/stats_mesh.txt | tail -45 | head -20
timeout 400 python bench.py --config 4 --steps 2 --warmup 1 --chain 1 --no-extras --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/stats_c4.txt
echo "== c4"; grep "\[stats\]" $OUT/stats_c4.txt | tail -45 | head -20
