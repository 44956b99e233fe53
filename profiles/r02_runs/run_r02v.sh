#!/bin/bash
# round-2 pass v: do sparse waves (few live lanes) finish sooner with every stage threshold at 1?  Single launches (their tail) and the 8-way tile slice
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02v
rm -rf $OUT; mkdir -p $OUT
D=$REPO/raytracing-in-one-weekend_amd/csrc/build
for rep in 1 2; do
  for v in product sparse16 sparse32; do
    if [ $v = product ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$D/librtow_hip_$v.so; fi
    echo -n "$v plain: "; timeout 200 python bench.py --steps 12 --warmup 1 --chain 1 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernel_ms_per_step'])"
    echo -n "$v chain: "; timeout 200 python bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernel_ms_per_step'])"
  done
done
for v in product sparse16 sparse32; do
  if [ $v = product ]; then unset RTOW_LIB_PATH; else export RTOW_LIB_PATH=$D/librtow_hip_$v.so; fi
  echo -n "$v tiles C2 8 slices: "; timeout 300 python profiles/emulate_tile_split.py --config 2 --slices 1,8 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({g:(v['slowest_ms'],v['predicted_speedup_kernel']) for g,v in d['slices'].items()})"
done
