#!/bin/bash
# round-2 pass t: final build - the driver's bench command for all four single-GPU configs (complete JSON lines), profile r02c
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02t
rm -rf $OUT; mkdir -p $OUT
timeout 300 python bench.py --steps 20 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
for c in 3 4 5; do timeout 400 python bench.py --config $c --steps 8 --warmup 1 > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; done
bash profiles/collect.sh r02c > /dev/null 2>&1
for c in 2 3 4 5; do python -c "import json; d=json.load(open('$OUT/bench_c$c.json')); print($c, d['value'], d['ms_per_step'], d['config']['batches_per_launch'], d.get('plain_batches',{}).get('value'), d.get('host_buffer_ms_per_step'), d.get('host_buffer_chain_ms_per_step'), d['cpu_baseline']['value'], d['mrays_per_s'])"; done
