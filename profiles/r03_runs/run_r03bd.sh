#!/bin/bash
# round-3 pass bd: the build with the hand-over at 3 candidates and the walk slices re-set: GPU suite, the driver's command, the other configs, a last slice check
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bd
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/gpu_suite.log 2>&1; tail -3 $OUT/gpu_suite.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python -c "
import json; d=json.load(open('$OUT/bench_driver.json')); print('driver', d['value'], d['ms_per_step'], d['config']['threshold_set'], d['config']['scheduler_tune'], d['plain_batches']['value'], d['chain2']['value'], d['mrays_per_s'])"
run() { timeout 400 python bench.py $1 --steps $2 --warmup 2 --no-extras --no-cpu-baseline $4 > $OUT/$3.json 2>$OUT/err.log; python -c "
import json; d=json.load(open('$OUT/$3.json')); print('$3', d['value'], d['kernel_ms_per_step'], d['config']['threshold_set'], d['config']['scheduler_tune'])" || tail -2 $OUT/err.log; }
for rep in 1 2; do
run "--config 3" 2 c3_$rep; run "--config 4" 8 c4_$rep; run "--config 5" 8 c5_$rep; run "--scene mesh" 8 mesh_$rep
run "--scene mesh" 8 mesh_slice20_$rep "--tune 16,48,1,1,1,1,3,1,20"; run "--scene mesh" 8 mesh_slice28_$rep "--tune 16,48,1,1,1,1,3,1,28"
run "--config 4" 8 c4_slice12_$rep "--tune 24,32,1,32,28,1,3,1,12"; run "--config 4" 8 c4_slice14_$rep "--tune 24,32,1,32,28,1,3,1,14"
run "--scene mixed --spp 64" 8 mixed_$rep; run "--scene textured --spp 64" 8 textured_$rep; run "--scene volumes --spp 32" 8 volumes_$rep; run "--scene meshfog --spp 16" 4 meshfog_$rep
done
