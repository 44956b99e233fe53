#!/bin/bash
# round-3 pass by: final-state validation: build check, smoke, the driver's default and explicit commands, the GPU suite, the two-rank debug bench
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03by
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1]); print('default', d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline_c1']['value'], {k: v['GBps'] for k, v in d['post_passes']['3840x2160'].items()})"; grep real $OUT/bench_default.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python -c "
import json; d=json.load(open('$OUT/bench_driver.json')); print('driver', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_suite.log 2>&1; tail -3 $OUT/gpu_suite.log
