#!/bin/bash
# development helper: bench.py under several RTOW_TUNE settings (regen,trav,test,hit,sky,vol,-,slice: 8 integers, see rtow_api.hip)
for t in "$@"; do
  echo -n "$t : "
  RTOW_TUNE=$t python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernel_ms_per_step'])"
done
