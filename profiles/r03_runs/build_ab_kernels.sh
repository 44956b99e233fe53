#!/bin/bash
# A/B build of the small-kernel unit only: compiles rtow_kernels.hip with the given -D flags and links it against the product's other objects.
#   profiles/r03_runs/build_ab_kernels.sh <tag> <flags...>   ->  raytracing-in-one-weekend_amd/csrc/build/librtow_hip_<tag>.so   (select with RTOW_LIB_PATH)
set -e
cd "$(dirname "$0")/../../raytracing-in-one-weekend_amd/csrc"
tag=$1; shift
mkdir -p build/ab_$tag
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math "$@" --offload-arch=gfx950 -x hip -c rtow_kernels.hip -o build/ab_$tag/rtow_kernels.o
others=$(ls build/*.o | grep -v "build/rtow_kernels.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/librtow_hip_$tag.so build/ab_$tag/rtow_kernels.o $others -ldl
echo build/librtow_hip_$tag.so
