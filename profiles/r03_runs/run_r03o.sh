#!/bin/bash
# round-3 pass o: camera-ray list loads as non-temporal (streaming) loads - does it keep half-written accumulator lines in L2 longer (HBM write traffic), at what speed?
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03o
rm -rf $OUT; mkdir -p $OUT
NT=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_nt.so
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2 3; do
  timeout 200 python bench.py $ARGS > $OUT/bench_product_$rep.json 2>> $OUT/bench.err
  RTOW_LIB_PATH=$NT timeout 200 python bench.py $ARGS > $OUT/bench_nt_$rep.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
ONE="python $REPO/bench.py --steps 10 --warmup 0 --chain 10 --no-cpu-baseline --no-extras"
for lib in product nt; do for c in FETCH_SIZE WRITE_SIZE; do
  if [ $lib = nt ]; then export RTOW_LIB_PATH=$NT; else unset RTOW_LIB_PATH; fi
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${lib}_$c -o bench -- $ONE > $OUT/pmc_${lib}_$c.log 2>&1
done; done
python - <<PY
import csv,glob
for lib in ("product","nt"):
  for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("$OUT/pmc_%s_%s/**/*counter_collection.csv"%(lib,c), recursive=True)
    rows=[r for r in csv.DictReader(open(f[0])) if "sample_batch_kernel" in r["Kernel_Name"]]
    last=max(int(r["Dispatch_Id"]) for r in rows)
    print(lib, c, sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"])==last)*1024/1e9, "GB raw")
PY
