#!/bin/bash
# round-3 pass bj: after the last host-side edit (tuner skips captured streams, event handles): API / threshold / chain tests and the driver's command
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bj
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_chain.py tests/test_gpu_fullsize.py -q -x -k "not config3" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python -c "
import json; d=json.load(open('$OUT/bench_driver.json')); print('driver', d['value'], d['ms_per_step'], d['config']['threshold_set'])"
