#!/bin/bash
# round-3 pass az: per-scene threshold measurement built in: its test, which set each scene gets and what it is worth, the driver's command
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03az
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "thresholds or schedule" > $OUT/test.log 2>&1; tail -3 $OUT/test.log
run() { timeout 400 python bench.py $1 --steps $2 --warmup 2 --no-extras --no-cpu-baseline $4 > $OUT/$3.json 2>$OUT/err.log; python -c "
import json; d=json.load(open('$OUT/$3.json')); print('$3', d['value'], d['kernel_ms_per_step'], d['config']['threshold_set'], d['config']['scheduler_tune'])" || tail -2 $OUT/err.log; }
for rep in 1 2; do
run "" 20 cover_tuned_$rep; run "" 20 cover_builtin_$rep "--context-flags 64"
run "--config 4" 8 c4_tuned_$rep; run "--config 4" 8 c4_builtin_$rep "--context-flags 64"
run "--config 5" 8 c5_tuned_$rep; run "--config 5" 8 c5_builtin_$rep "--context-flags 64"
run "--scene mesh" 8 mesh_tuned_$rep; run "--scene mesh" 8 mesh_builtin_$rep "--context-flags 64"
run "--scene mixed --spp 64" 8 mixed_tuned_$rep; run "--scene mixed --spp 64" 8 mixed_builtin_$rep "--context-flags 64"
run "--scene textured --spp 64" 8 textured_tuned_$rep; run "--scene textured --spp 64" 8 textured_builtin_$rep "--context-flags 64"
run "--scene volumes --spp 32" 8 volumes_tuned_$rep; run "--scene volumes --spp 32" 8 volumes_builtin_$rep "--context-flags 64"
run "--scene meshfog --spp 16" 4 meshfog_tuned_$rep; run "--scene meshfog --spp 16" 4 meshfog_builtin_$rep "--context-flags 64"
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python -c "
import json; d=json.load(open('$OUT/bench_driver.json')); print('driver', d['value'], d['ms_per_step'], d['config']['threshold_set'])"
