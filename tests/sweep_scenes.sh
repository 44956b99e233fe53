#!/bin/bash
# development helper: kernel time of several scenes under scheduler settings given as arguments
# (RtowContextOptions.schedulerTune: regen,trav,test,hit,sky,vol,-,-,slice = 9 integers)
for t in "$@"; do
  for sc in cover moving stress mixed; do
    echo -n "$t $sc : "
    python tests/run_gpu_quick.py 1920 1080 64 8 $sc $t 2>&1 | grep "iter 2" | sed 's/.*kernel //'
  done
done
