#!/bin/bash
# round-2 pass r05h: the whole library built with -mllvm -amdgpu-sched-strategy=max-ilp against the shipped build, alternating runs on one box
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$1', d['value'], d['kernel_ms_per_step'])"; }
for i in 1 2 3; do
  python bench.py --steps 16 --warmup 2 --no-extras 2>/dev/null | line base
  RTOW_LIB_PATH=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_ilp.so python bench.py --steps 16 --warmup 2 --no-extras 2>/dev/null | line maxilp
done
