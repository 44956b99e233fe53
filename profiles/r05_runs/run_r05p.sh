# round 5, GPU call p: one chunk list per image region / XCD (schedulerTune[7] + 256 x 8) against the device-wide list: parity, then same-box A/B on mesh, C2, C4, C5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
python -m pytest tests/test_gpu_regroup.py tests/test_gpu_chain.py -q -x -n 4 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
for R in 1 2; do for V in 3 2051; do
  T=0,0,0,0,0,0,0,$V,0
  python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras --tune $T > $O/mesh_${V}_$R.json 2> $O/mesh_${V}_$R.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune $T > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  for C in 4 5; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $T > $O/c${C}_${V}_$R.json 2> $O/c${C}_${V}_$R.err; done
done; done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05p/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d["value"])
        if d.get("roofline", {}).get("traffic"): res[name + " traffic"].append(d["roofline"]["traffic"])
        for k in ("plain_batches", "chain2", "group_fold"):
            if k in d: res[name + " " + k].append(d[k]["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
