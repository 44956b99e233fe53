#!/bin/bash
# round-2 third pass: chained batches (tests + timing), comm test diagnostics
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02c
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_chain.py -m gpu -x -q > $OUT/pytest_chain.log 2>&1; echo "rc=$?" >> $OUT/pytest_chain.log
NCCL_DEBUG=INFO timeout 300 python -m pytest tests/test_gpu_comm.py -m gpu -q -rs > $OUT/pytest_comm.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_plain.json 2> $OUT/bench_plain.err
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest_chain.log; tail -30 $OUT/pytest_comm.log | cut -c1-300; cat $OUT/bench_plain.json | cut -c1-200; tail -3 $OUT/pytest.log
