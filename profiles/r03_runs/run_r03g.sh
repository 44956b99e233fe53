#!/bin/bash
# round-3 pass g: stage thresholds for the mesh scene (general-entity kernels).  The stats build (r03f) shows its exact tests running with 12 of 64
# lanes and its shading with 14: does holding those stages back until more lanes wait pay on THIS kind of scene (it never did on spheres)?
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03g
rm -rf $OUT; mkdir -p $OUT
run() { timeout 300 python bench.py --scene mesh --steps 4 --warmup 1 --no-extras --no-cpu-baseline --tune $1 > $OUT/mesh_$1.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/mesh_$1.json')); print('$1', d['value'], d['kernel_ms_per_step'])"; }
for T in 1 8 16 24 32; do for H in 1 8 16 24; do run 16,48,$T,$H,1,1,1,1,16; done; done
for S in 8 24 32; do run 16,48,1,1,1,1,1,1,$S; done
run 16,32,1,1,1,1,1,1,16
run 16,56,1,1,1,1,1,1,16
