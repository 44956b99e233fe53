#!/bin/bash
# round-3 pass k: chained launches with per-XCD chunk ownership (plain stores, sc1 loads) against the round-2 protocol (sc1 write-through stores): parity, then timing and HBM traffic
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03k
rm -rf $OUT; mkdir -p $OUT
PREV=$REPO/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_prev.so
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_matrix.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
timeout 600 python tests/soak_chain.py 0.5 > $OUT/soak_chain.log 2>&1; tail -3 $OUT/soak_chain.log
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2 3; do
  RTOW_LIB_PATH=$PREV timeout 200 python bench.py $ARGS > $OUT/bench_prev_$rep.json 2>> $OUT/bench.err
  timeout 200 python bench.py $ARGS > $OUT/bench_new_$rep.json 2>> $OUT/bench.err
done
for f in $OUT/bench_*.json; do echo -n "$(basename $f) "; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['kernel_ms_per_step'], d['mrays_per_s'])"; done
cd /tmp && export TMPDIR=/tmp
ONE="python $REPO/bench.py --steps 10 --warmup 0 --chain 10 --no-cpu-baseline --no-extras"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- $ONE > $OUT/pmc_$c.log 2>&1; done
python - <<PY
import csv,glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("$OUT/pmc_%s/**/*counter_collection.csv"%c, recursive=True)
    rows=[r for r in csv.DictReader(open(f[0])) if "sample_batch_kernel" in r["Kernel_Name"]]
    last=max(int(r["Dispatch_Id"]) for r in rows)
    print(c, sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"])==last)*1024/1e9, "GB (raw KiB -> bytes; read side x2 on gfx950)")
PY
