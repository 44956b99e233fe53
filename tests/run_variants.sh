#!/bin/bash
# development helper: each (kind, tree placement) of tests/test_gpu_variants.py in its own process under a timeout, so that a hanging
# kernel variant is named instead of stalling the whole run.  Log: gpurun_out/variants.log; prints the number of groups that passed (of 20)
mkdir -p gpurun_out
: > gpurun_out/variants.log
for kind in spheres spheres_ties spheres_motion spheres_motion_ties general general_ties volumes textured textured_ties volumes_textured; do
  for lds in lds hbm; do
    echo "== $kind-$lds" >> gpurun_out/variants.log
    timeout -k 5 ${RTOW_VARIANT_TIMEOUT:-90} python -m pytest "tests/test_gpu_variants.py::test_every_kernel_variant[$kind-$lds]" -q -m gpu -x 2>&1 | grep -v "^$" | tail -4 >> gpurun_out/variants.log
    echo "rc=${PIPESTATUS[0]}" >> gpurun_out/variants.log
  done
done
grep -c "^rc=0" gpurun_out/variants.log
