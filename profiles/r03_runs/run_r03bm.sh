#!/bin/bash
# round-3 pass bm: clear glass (roughness 0) advances the generator instead of drawing a direction it multiplies by zero: parity subset, A/B against the build before
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bm
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
run() { if [ $1 = new ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$1.so; fi
  timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2 3; do for lib in new base; do run $lib $rep "" 20 cover; done; done
for rep in 1 2; do for lib in new base; do run $lib $rep "--config 4" 8 c4; run $lib $rep "--config 5" 8 c5; run $lib $rep "--config 3" 2 c3; done; done
