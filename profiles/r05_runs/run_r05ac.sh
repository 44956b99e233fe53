# round 5, GPU call ac: the final build (slot tickets for groups) through the whole -m gpu suite and smoke; final driver-command line; counters of a group launch and of the host-default group launch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ac; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -4 > $O/tests_gpu.log; cat $O/tests_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05ac/bench_driver_command.json").read().strip().splitlines()[-1]); print("driver command", d["value"], d["ms_per_step"], {k: d[k]["value"] for k in ("plain_batches", "chain2", "group_fold", "plain_two_in_flight") if k in d}, {k: v["value"] for k, v in d["host_default"].items() if isinstance(v, dict) and "value" in v})
PY
POST=0 bash profiles/collect.sh r05_group 10 --only-leg group_fold > $O/collect_group.log 2>&1
POST=0 bash profiles/collect.sh r05_hostdefault 10 --only-leg host_default_group > $O/collect_hostdefault.log 2>&1
for f in $O/collect_*.log; do echo $f; tail -n 2 $f; done
