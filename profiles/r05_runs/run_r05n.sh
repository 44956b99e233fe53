# round 5, GPU call n: chained launches pixel by pixel (every lane carries its pixel through all batches) against batch by batch (RTOW_CONTEXT_BATCH_MAJOR_CHAINS) and against the build before the change
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
python -m pytest tests/test_gpu_chain.py tests/test_gpu_ties.py tests/test_gpu_regroup.py tests/test_gpu_golden.py -q -x -n 4 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
HEADLIB=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_head.so
for R in 1 2 3; do for V in pixel batch head; do
  F=0; unset RTOW_LIB_PATH
  [ $V = batch ] && F=256
  [ $V = head ] && export RTOW_LIB_PATH=$HEADLIB
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --context-flags $F > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  for C in 3 4 5; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras --context-flags $F > $O/c${C}_${V}_$R.json 2> $O/c${C}_${V}_$R.err; done
  python bench.py --steps 16 --warmup 4 --chain 4 --no-cpu-baseline --no-extras --context-flags $F > $O/c2chain4_${V}_$R.json 2> $O/c2chain4_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05n/c*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d["value"])
        for k in ("plain_batches", "chain2", "group_fold"):
            if k in d: res[name + " " + k].append(d[k]["value"])
        if "host_default" in d:
            for k in ("chain", "group_fold", "adaptive"): res[name + " host_default." + k].append(d["host_default"][k]["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
