# round 5, GPU call k: a tile's tickets most expensive first (schedulerTune[7] = 3): parity, then the driver's command with extras against the tiles in natural ticket order, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
python -m pytest tests/test_gpu_regroup.py -q -x -n 4 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
for R in 1 2 3; do for T in 1 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune 0,0,0,0,0,0,0,$T,0 > $O/c2_tune${T}_$R.json 2> $O/c2_tune${T}_$R.err
  for C in 4 5; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune 0,0,0,0,0,0,0,$T,0 > $O/c${C}_tune${T}_$R.json 2> $O/c${C}_tune${T}_$R.err; done
  for C in 2 4 5; do python bench.py --config $C --steps 10 --warmup 3 --chain 1 --no-cpu-baseline --no-extras --tune 0,0,0,0,0,0,0,$T,0 > $O/plain_c${C}_tune${T}_$R.json 2> $O/plain_c${C}_tune${T}_$R.err; done
done; done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05k/*_tune*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f)[:-7]
        res[name].append(d["value"])
        for k in ("plain_batches", "chain2", "plain_two_in_flight", "group_fold"):
            if k in d: res[name + " " + k].append(d[k]["value"])
        if "host_default" in d:
            for k in ("chain", "group_fold", "adaptive"): res[name + " host_default." + k].append(d["host_default"][k]["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-48s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
