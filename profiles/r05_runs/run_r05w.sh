# round 5, GPU call w: the cubemap's launch constants read on use (kernarg segment) instead of held in registers through every stage - against the build before (head), same box, alternating; parity of the cubemap / sky tests first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05w; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_matrix.py tests/test_gpu_golden.py -q -x -n 4 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
HEAD=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_head.so
for R in 1 2 3; do for V in new head; do
  unset RTOW_LIB_PATH; [ $V = head ] && export RTOW_LIB_PATH=$HEAD
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  for C in 4 5; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c${C}_${V}_$R.json 2> $O/c${C}_${V}_$R.err; done
  python bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/c3_${V}_$R.json 2> $O/c3_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05w/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
