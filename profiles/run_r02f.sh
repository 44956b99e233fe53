#!/bin/bash
# round-2 pass f: 144-variant build: variant sweep + full suite; stage statistics (stats build) of the headline workload
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02f
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so timeout 300 python tests/run_gpu_quick.py 1920 1080 256 8 cover > $OUT/stats_cover.log 2>&1
RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so timeout 300 python tests/run_gpu_quick.py 1920 1080 256 8 moving > $OUT/stats_moving.log 2>&1
timeout 300 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
tail -4 $OUT/pytest.log | cut -c1-300; grep "stats\]" $OUT/stats_cover.log | tail -60 | cut -c1-200; cut -c1-300 $OUT/bench_c2.json
