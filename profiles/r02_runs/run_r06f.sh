#!/bin/bash
# round-2 pass r06f: sphere normals through rtow::exact_div3 (three quotients on one correctly rounded reciprocal) against the same build with
# the compiler's three IEEE divisions (-DRTOW_EXACT_DIV3=0), alternating runs on one box
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$1', d['value'], d['kernel_ms_per_step'])"; }
for i in 1 2 3; do
  RTOW_LIB_PATH=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_nodiv3.so python bench.py --steps 16 --warmup 2 --no-extras 2>/dev/null | line ieee
  python bench.py --steps 16 --warmup 2 --no-extras 2>/dev/null | line div3
done
