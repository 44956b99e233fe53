#!/bin/bash
# round-2 pass r05b: final build (exact tests at wave priority 1, tie update as selects): the driver's bench command for the four single-GPU
# configs, the profile r02e (superseded by r02f, collected with profiles/collect.sh on the last build), the 2.5 x soak and both fuzz soaks
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05b
rm -rf $OUT; mkdir -p $OUT
timeout 300 python bench.py --steps 20 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
for c in 3 4 5; do timeout 400 python bench.py --config $c --steps 8 --warmup 1 > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; done
bash profiles/collect.sh r02e > /dev/null 2>&1
for c in 2 3 4 5; do python -c "import json; d=json.loads([l for l in open('$OUT/bench_c$c.json') if l.startswith(chr(123))][-1]); print($c, d['value'], d['ms_per_step'], d['config']['batches_per_launch'], d.get('plain_batches',{}).get('value'), d.get('host_buffer_ms_per_step'), d.get('host_buffer_chain_ms_per_step'), d['cpu_baseline']['value'], d['mrays_per_s'])"; done
timeout 900 python tests/soak_frames.py 2.5 > $OUT/soak.log 2>&1; tail -1 $OUT/soak.log
RTOW_FUZZ_SEEDS=8000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1
RTOW_FUZZ_HEAVY=1 RTOW_FUZZ_SEEDS=4000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1
