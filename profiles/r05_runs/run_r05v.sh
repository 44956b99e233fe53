# round 5, GPU call v: queue slots reserved K at a time per wave (schedulerTune[7] = 3 + 256 K; the chunks a wave works through one after the other are neighbours in the cost order): parity on chains / groups, then same-box A/B, K = 1 2 4 8
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
python -m pytest tests/test_gpu_chain.py tests/test_gpu_regroup.py -q -x -n 4 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
for R in 1 2; do for K in 1 2 4 8; do
  T=0,0,0,0,0,0,0,$((3 + 256 * K)),0
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune $T > $O/c2_K${K}_$R.json 2> $O/c2_K${K}_$R.err
  python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $T > $O/c5_K${K}_$R.json 2> $O/c5_K${K}_$R.err
  python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $T > $O/c4_K${K}_$R.json 2> $O/c4_K${K}_$R.err
done; done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05v/*_*_*.json")):
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
