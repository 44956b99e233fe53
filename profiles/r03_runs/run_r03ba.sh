#!/bin/bash
# round-3 pass ba: the per-sample policies (units = 16-sample groups; no threshold measurement there) under the old and the new built-in thresholds
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ba
rm -rf $OUT; mkdir -p $OUT
run() { timeout 400 python bench.py --rng per-sample $3 --steps $4 --warmup 2 --no-extras --no-cpu-baseline $1 > $OUT/$5_$2.json 2>$OUT/err.log; python -c "
import json; d=json.load(open('$OUT/$5_$2.json')); print('$5', d['value'], d['kernel_ms_per_step'], d['config']['threshold_set'], d['config']['scheduler_tune'])" || tail -2 $OUT/err.log; }
for rep in 1 2; do
run "" $rep "" 8 cover_builtin; run "--tune 16,48,1,1,28,1,1,1,16" $rep "" 8 cover_old; run "--tune 16,48,1,1,1,1,1,1,16" $rep "" 8 cover_general; run "--tune 24,32,1,16,28,1,1,1,16" $rep "" 8 cover_hit16
run "" $rep "--config 5" 4 c5_builtin; run "--tune 16,48,1,1,28,1,1,1,16" $rep "--config 5" 4 c5_old
done
