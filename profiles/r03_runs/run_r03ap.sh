#!/bin/bash
# round-3 pass ap: new default thresholds built in: walk-slice sweep around them, the driver's bench command, and the GPU suite
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ap
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline $1 > $OUT/$5_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$2.json')); print('$5', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2; do
  run "" $rep "" 20 cover_default
  for s in 12 14 18 20; do run "--tune 24,32,1,32,28,1,1,1,$s" $rep "" 20 cover_slice$s; done
  run "" $rep "--config 4" 8 c4_default
  for s in 12 20 24; do run "--tune 24,32,1,32,28,1,1,1,$s" $rep "--config 4" 8 c4_slice$s; done
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python -c "
import json; d=json.load(open('$OUT/bench_driver.json')); print('driver', d['value'], d['ms_per_step'], d['roofline'], d.get('plain_batches'), d.get('chain2'))"
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_suite.log 2>&1; tail -3 $OUT/gpu_suite.log
