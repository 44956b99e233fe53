#!/bin/bash
# round-2 passes r04m, r04n: wave priority per stage (s_setprio, -DRTOW_EXPERIMENT_STAGE_PRIO=<2 bits per stage>) against the shipped build,
# alternating runs on one box.  r04m: 64 = HIT high, 4 = walk high, 80 = exact tests + HIT, 16 = exact tests; r04n: 48 = tests at 3, 17 = + REGEN, 272 = + SKY, 16400 = + scheduler, 336 = tests + HIT + SKY
B=raytracing-in-one-weekend_amd/csrc/build
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$1', d['value'], d['kernel_ms_per_step'])"; }
for i in 1 2 3; do
  python bench.py --steps 16 --warmup 2 --no-extras 2>/dev/null | line base
  for v in 16 48 17 272 16400 336; do RTOW_LIB_PATH=$B/librtow_hip_p$v.so python bench.py --steps 16 --warmup 2 --no-extras 2>/dev/null | line prio$v; done
done
