#!/bin/bash
# round-2 pass p: what do the exact-tie kernels cost now (RTOW_CONTEXT_EXACT_TIES_ALWAYS), same box, alternating runs; cover + mixed-primitive scene
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02p
rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  timeout 200 python bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_default_$rep.json 2>> $OUT/bench.err
  timeout 200 python bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-extras --context-flags 1 > $OUT/bench_exactties_$rep.json 2>> $OUT/bench.err
done
for c in 4 5; do
  timeout 200 python bench.py --config $c --steps 8 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_default_c$c.json 2>> $OUT/bench.err
  timeout 200 python bench.py --config $c --steps 8 --warmup 1 --no-cpu-baseline --no-extras --context-flags 1 > $OUT/bench_exactties_c$c.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'])"; done
