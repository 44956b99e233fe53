# round 5, GPU call h: sphere roots divided through the ray's own reciprocal (exact_div_by) against the compiler's IEEE division: parity, then same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_detmath.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py -q -x -n 4 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
OFF=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_nodiv.so
for R in 1 2 3; do for L in new old; do for C in 2 4 5; do
  if [ $L = old ]; then export RTOW_LIB_PATH=$OFF; else unset RTOW_LIB_PATH; fi
  python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c${C}_${L}_$R.json 2> $O/c${C}_${L}_$R.err
done; done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05h/c*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c, l, r = os.path.basename(f)[:-5].split("_"); res[(c, l)].append(d["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print(k, res[k], sum(res[k]) / len(res[k]))
PY
