#!/bin/bash
# round-3 pass bw: third threshold family in the per-scene measurement: its test, which set each scene gets and what it is worth
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bw
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_abi.py -q -x -k "thresholds or schedule or abi or mirror" > $OUT/test.log 2>&1; tail -3 $OUT/test.log
run() { timeout 400 python bench.py $1 --steps $2 --warmup 2 --no-extras --no-cpu-baseline $4 > $OUT/$3.json 2>$OUT/err.log; python -c "
import json; d=json.load(open('$OUT/$3.json')); print('$3', d['value'], d['kernel_ms_per_step'], d['config']['threshold_set'], d['config']['scheduler_tune'])" || tail -2 $OUT/err.log; }
for rep in 1 2; do
run "" 20 cover_$rep; run "--config 4" 8 c4_$rep; run "--config 5" 8 c5_$rep; run "--scene mesh" 8 mesh_$rep
run "--scene mixed --spp 64" 8 mixed_$rep; run "--scene textured --spp 64" 8 textured_$rep; run "--scene volumes --spp 32" 8 volumes_$rep; run "--scene meshfog --spp 16" 4 meshfog_$rep
done
