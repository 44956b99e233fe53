#!/bin/bash
# round-2 pass e: chain table in device memory; api tests in isolation; bench at chain 8 / 16; stats build for the stage profile
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02e
rm -rf $OUT; mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_chain.py -m gpu -x -q > $OUT/pytest_chain.log 2>&1; rc=$?; echo "rc=$rc" >> $OUT/pytest_chain.log
if [ $rc -eq 124 ]; then echo "chain tests hung - stopping here" ; tail -5 $OUT/pytest_chain.log; exit 0; fi
timeout 400 python -m pytest tests/test_gpu_api.py -m gpu -x -q > $OUT/pytest_api.log 2>&1; echo "rc=$?" >> $OUT/pytest_api.log
for ch in 8 16 1; do timeout 300 python bench.py --steps 16 --warmup 2 --chain $ch --no-cpu-baseline --no-extras > $OUT/bench_chain$ch.json 2> $OUT/bench_chain$ch.err; done
tail -3 $OUT/pytest_chain.log; tail -12 $OUT/pytest_api.log | cut -c1-300; for ch in 8 16 1; do cut -c1-220 $OUT/bench_chain$ch.json; done
