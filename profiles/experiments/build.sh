#!/bin/bash
# Builds a development variant of librtow_hip.so from a COPY of raytracing-in-one-weekend_amd/csrc with the instrumentation patch applied: the stage
# statistics (RTOW_STATS), the timing experiments of HISTORY.md (RTOW_EXPERIMENT_DOUBLE_WALK / DOUBLE_TEST / ALL_LAMBERT / COHERENT_WAVES / SCATTER_TICKETS,
# RTOW_FINALIZE_EXPERIMENT) and the ballot / prefix-sum compaction of the exact tests (RTOW_COMPACT_TESTS=1) live in that patch and nowhere in the shipped sources.
#   bash profiles/experiments/build.sh <name> [compiler flags ...]        e.g.   build.sh stats -DRTOW_STATS      build.sh compact -DRTOW_COMPACT_TESTS=1
# -> raytracing-in-one-weekend_amd/csrc/build/librtow_hip_<name>.so ; run anything against it with RTOW_LIB_PATH=<that file> (the loader's development override).
# A/B runs compare it with a same-box run of the product library; nothing built here is ever shipped.
set -e
NAME=${1:?usage: build.sh <name> [flags]}
shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
WORK=$(mktemp -d)
mkdir -p "$WORK/raytracing-in-one-weekend_amd" "$WORK/include"
cp -r "$ROOT/raytracing-in-one-weekend_amd/csrc" "$WORK/raytracing-in-one-weekend_amd/csrc"
cp "$ROOT/include/rtow.h" "$WORK/include/rtow.h"
rm -rf "$WORK/raytracing-in-one-weekend_amd/csrc/build" "$WORK"/raytracing-in-one-weekend_amd/csrc/*.so
(cd "$WORK/raytracing-in-one-weekend_amd" && patch -p1 < "$ROOT/profiles/experiments/instrumentation.patch")
make -C "$WORK/raytracing-in-one-weekend_amd/csrc" EXTRA="$*" LIB=librtow_hip_$NAME.so
mkdir -p "$ROOT/raytracing-in-one-weekend_amd/csrc/build"
cp "$WORK/raytracing-in-one-weekend_amd/csrc/librtow_hip_$NAME.so" "$ROOT/raytracing-in-one-weekend_amd/csrc/build/"
rm -rf "$WORK"
echo "built raytracing-in-one-weekend_amd/csrc/build/librtow_hip_$NAME.so"
