#!/bin/bash
# A/B build of ONE translation unit: compiles <unit>.hip with the given -D flags and links it against the product's other objects.
#   profiles/r03_runs/build_ab_unit.sh <tag> <unit> <flags...>   ->  raytracing-in-one-weekend_amd/csrc/build/librtow_hip_<tag>.so   (select with RTOW_LIB_PATH)
set -e
cd "$(dirname "$0")/../../raytracing-in-one-weekend_amd/csrc"
tag=$1; unit=$2; shift; shift
mkdir -p build/ab_$tag
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math "$@" --offload-arch=gfx950 -x hip -c $unit.hip -o build/ab_$tag/$unit.o
others=$(ls build/*.o | grep -v "build/$unit.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/librtow_hip_$tag.so build/ab_$tag/$unit.o $others -ldl
echo build/librtow_hip_$tag.so
