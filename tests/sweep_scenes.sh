#!/bin/bash
# development helper: kernel time of several scenes under RTOW_TUNE settings given as arguments
for t in "$@"; do
  for sc in cover moving stress mixed; do
    echo -n "$t $sc : "
    RTOW_TUNE=$t python tests/run_gpu_quick.py 1920 1080 64 8 $sc 2>&1 | grep "iter 2" | sed 's/.*kernel //'
  done
done
