# round 5, GPU call y: view / sky constants from an LDS copy in every kernel (lds, RTOW_LDS_VIEW=3) against the shipped mix (new: registers where the scene is in LDS, read on use elsewhere) and the cold-cubemap-only build (cc)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05y; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x -n 4 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
B=raytracing-in-one-weekend_amd/csrc/build
for R in 1 2 3; do for V in new lds cc; do
  unset RTOW_LIB_PATH; [ $V != new ] && export RTOW_LIB_PATH=$B/librtow_hip_$V.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  for C in 4 5; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c${C}_${V}_$R.json 2> $O/c${C}_${V}_$R.err; done
  python bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/c3_${V}_$R.json 2> $O/c3_${V}_$R.err
  python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/mesh_${V}_$R.json 2> $O/mesh_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05y/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d["value"])
        for k in ("plain_batches", "chain2", "group_fold"):
            if k in d: res[name + " " + k].append(d[k]["value"])
        if "host_default" in d:
            for k in ("chain", "group_fold", "adaptive"):
                if k in d["host_default"]: res[name + " host_default." + k].append(d["host_default"][k]["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
