#!/bin/bash
# round-2 pass d: chained batches after the livelock fix (tight timeouts: a hang must not hold the box), reference diagnostics, full suite, bench
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02d
rm -rf $OUT; mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_chain.py -m gpu -x -q > $OUT/pytest_chain.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest_chain.log
if [ $rc -eq 124 ]; then echo "chain tests hung - stopping here" ; tail -5 $OUT/pytest_chain.log; exit 0; fi
timeout 300 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "reference_identical or registered or slice_that or cancelled_per or overflow" > $OUT/pytest_api.log 2>&1; echo "rc=$?" >> $OUT/pytest_api.log
timeout 300 python bench.py --steps 16 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
for c in 4 5; do timeout 300 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; done
tail -4 $OUT/pytest_chain.log; tail -15 $OUT/pytest_api.log | cut -c1-300; cat $OUT/bench_c2.json | cut -c1-1500; tail -5 $OUT/bench_c2.err; tail -12 $OUT/pytest.log | cut -c1-300
