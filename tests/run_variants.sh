#!/bin/bash
# development helper: each (kind, tree placement) of tests/test_gpu_variants.py in its own process under a timeout, so that a hanging
# kernel variant is named instead of stalling the whole run.  Log: gpurun_out/variants.log
mkdir -p gpurun_out
: > gpurun_out/variants.log
for kind in spheres spheres_motion general volumes textured volumes_textured; do
  for lds in lds hbm; do
    echo "== $kind-$lds" >> gpurun_out/variants.log
    timeout -k 5 ${RTOW_VARIANT_TIMEOUT:-90} python -m pytest tests/test_gpu_variants.py -q -m gpu -x -k "$lds and $kind" 2>&1 | grep -v "^$" | tail -4 >> gpurun_out/variants.log
    echo "rc=${PIPESTATUS[0]}" >> gpurun_out/variants.log
  done
done
grep -c "1 passed" gpurun_out/variants.log
