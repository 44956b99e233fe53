#!/bin/bash
# round-2 pass l: scheduler threshold sweep on the final build (chains of 8), chain length 16
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02l
rm -rf $OUT; mkdir -p $OUT
run() { timeout 120 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernel_ms_per_step'])"; }
echo -n "default(16,48,1,1,1,1,1,1,16) : "; run | tee -a $OUT/sweep.log
for t in 16,40,1,1,1,1,1,1,16 16,56,1,1,1,1,1,1,16 16,32,1,1,1,1,1,1,16 8,48,1,1,1,1,1,1,16 24,48,1,1,1,1,1,1,16 32,48,1,1,1,1,1,1,16 16,48,1,1,1,1,1,1,12 16,48,1,1,1,1,1,1,20 16,48,1,1,1,1,1,1,24 16,48,4,1,1,1,1,1,16 16,48,1,4,1,1,1,1,16 16,48,1,1,4,1,1,1,16 16,48,1,1,8,1,1,1,16 24,56,1,1,1,1,1,1,20; do
  echo -n "$t : " | tee -a $OUT/sweep.log; run --tune $t | tee -a $OUT/sweep.log
done
echo -n "default again : "; run | tee -a $OUT/sweep.log
echo -n "chain 16 : "; run --chain 16 | tee -a $OUT/sweep.log
