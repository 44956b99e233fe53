#!/bin/bash
# round-2 pass i: exact_sqrt with the restricted range (exhaustive parity), time-fraction hoist for moving spheres, full suite, A/B
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02i
rm -rf $OUT; mkdir -p $OUT
timeout 300 ./tests/build/exactmath_parity > $OUT/parity.log 2>&1; echo "rc=$?" >> $OUT/parity.log
cat $OUT/parity.log
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_noexact.so
for rep in 1 2; do
  timeout 200 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_exact_$rep.json 2>> $OUT/bench.err
  RTOW_LIB_PATH=$B timeout 200 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_noexact_$rep.json 2>> $OUT/bench.err
done
for c in 4 5 3; do
  timeout 300 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_exact_c$c.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'])"; done
