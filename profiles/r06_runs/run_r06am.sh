# round 6, GPU call am: the tree as committed (library rebuilt from it): the GPU suite as the driver runs it, smoke, the driver's command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06am; mkdir -p $O
( time timeout 1800 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu_serial.log 2>&1; tail -5 $O/pytest_gpu_serial.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06am/bench_driver_command.json").read().strip().splitlines()[-1])
print("driver", d["value"], "per_sample", d.get("per_sample", {}).get("value"), "host_default", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("host_default", {}).items() if k in ("chain", "group_fold", "adaptive")})
PY
