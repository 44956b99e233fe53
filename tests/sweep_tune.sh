#!/bin/bash
# development helper: bench.py under several scheduler settings (RtowContextOptions.schedulerTune: regen,trav,test,hit,sky,vol,-,-,slice = 9 integers)
for t in "$@"; do
  echo -n "$t : "
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernel_ms_per_step'])"
done
