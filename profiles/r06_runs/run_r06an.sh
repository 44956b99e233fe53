# round 6, GPU call an: a launch that carries a rate no twin serves is refused (launchByDiagGeo) - the GPU suite goes through every variant and must not meet one; smoke; the driver's command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06an; mkdir -p $O
( time timeout 1800 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu_serial.log 2>&1; tail -5 $O/pytest_gpu_serial.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06an/bench_driver_command.json").read().strip().splitlines()[-1])
print("driver", d["value"], "per_sample", d.get("per_sample", {}).get("value"), "host_default", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("host_default", {}).items() if k in ("chain", "group_fold", "adaptive")})
PY
timeout 600 python tests/soak_chain.py 0.5 2>&1 | tail -1
