#!/bin/bash
# round-3 pass aw: waves that hold the launch's most expensive chunks run at wave priority 3 (tune[7] - 1 = the hot share of the chunks in 64ths): slices and whole frames
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03aw
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py -q -x -k "cover or tiny or slices or chain" > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
for h in 1 5 9 17 33; do
  t=24,32,1,32,28,1,1,$h,16
  timeout 600 python profiles/emulate_tile_split.py --config 2 --slices 1,2,4,8 --tune $t > $OUT/tiles_c2_$h.json 2> $OUT/err_c2_$h.log
  python -c "
import json; d=json.load(open('$OUT/tiles_c2_$h.json')); print('hot $h c2', {k:(v['slowest_ms'], [round(x,1) for x in v['kernel_ms_per_slice']]) for k,v in d['slices'].items()})"
done
run() { timeout 400 python bench.py $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline --tune $1 > $OUT/$5_$1_$2.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$5_$1_$2.json')); print('$5 $1', d['value'], d['kernel_ms_per_step'])"; }
for rep in 1 2; do for h in 1 5 9 17; do run 24,32,1,32,28,1,1,$h,16 $rep "" 20 cover; run 24,32,1,32,28,1,1,$h,16 $rep "--chain 1" 8 plain; done; done
