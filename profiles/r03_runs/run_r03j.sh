#!/bin/bash
# round-3 pass j: ballot / prefix-sum compaction of the exact tests (RTOW_COMPACT_TESTS=1 build) against the product: parity first, then timing
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03j
rm -rf $OUT; mkdir -p $OUT
B=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_compact.so
RTOW_LIB_PATH=$B timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "spheres or cover or golden or moving or slices or config" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
RTOW_LIB_PATH=$B timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config2 or config4 or config5 or schedule" > $OUT/tests_full.log 2>&1; tail -3 $OUT/tests_full.log
ARGS="--steps 16 --warmup 1 --no-cpu-baseline --no-extras"
for rep in 1 2 3; do
  timeout 200 python bench.py $ARGS > $OUT/bench_product_$rep.json 2>> $OUT/bench.err
  RTOW_LIB_PATH=$B timeout 200 python bench.py $ARGS > $OUT/bench_compact_$rep.json 2>> $OUT/bench.err
done
for c in 4 5; do
  timeout 200 python bench.py --config $c $ARGS > $OUT/bench_c${c}_product.json 2>> $OUT/bench.err
  RTOW_LIB_PATH=$B timeout 200 python bench.py --config $c $ARGS > $OUT/bench_c${c}_compact.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'], d['mrays_per_s'])"; done
