#!/bin/bash
# round-2 pass j: sampleAlbedo rebuilt from the path history (no VGPR spills), cold kernargs loaded at pixel boundaries (SGPR spills 81 -> 4)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02j
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2; do timeout 200 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_c2_$rep.json 2>> $OUT/bench.err; done
for c in 4 5 3; do timeout 300 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_c$c.json 2>> $OUT/bench.err; done
timeout 200 python bench.py --steps 16 --warmup 2 --chain 1 --no-cpu-baseline --no-extras > $OUT/bench_c2_plain.json 2>> $OUT/bench.err
timeout 200 python bench.py --steps 16 --warmup 2 --rng per-sample --no-cpu-baseline --no-extras > $OUT/bench_c2_persample.json 2>> $OUT/bench.err
for f in $OUT/bench_*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'])"; done
