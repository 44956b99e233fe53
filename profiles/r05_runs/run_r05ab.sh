# round 5, GPU call ab: batch groups with slot tickets (a wave reserves 4 (chunk, batch) slots at a time; K = 1 / 2 / 8 through schedulerTune[7] + 256 K): parity of everything that launches groups, then same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ab; mkdir -p $O
python -m pytest tests/test_gpu_group.py tests/test_gpu_chain.py tests/test_gpu_comm.py tests/test_gpu_regroup.py tests/test_gpu_ties.py tests/test_gpu_api.py -q -x -n 4 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
for R in 1 2 3; do for K in 4 1 2 8; do
  T=0,0,0,0,0,0,0,$((3 + 256 * K)),0
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-leg group_fold --tune $T > $O/group_K${K}_$R.json 2> $O/group_K${K}_$R.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-leg host_default_group --tune $T > $O/hostgroup_K${K}_$R.json 2> $O/hostgroup_K${K}_$R.err
done; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05ab/*_K*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d.get("value"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-300:])
for k in sorted(res): print("%-30s %s" % (k, res[k]))
d = json.loads(open("gpurun_out/r05ab/c2.json").read().strip().splitlines()[-1]); print("c2", d["value"], {k: d[k]["value"] for k in ("plain_batches", "chain2", "group_fold")}, {k: v["value"] for k, v in d["host_default"].items() if isinstance(v, dict) and "value" in v})
PY
