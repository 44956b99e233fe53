# round 6, GPU call p: the multi-workgroup chunk order with per-workgroup LDS counting (per-sample policies), mesh with the 16-visit slice; then the whole suite and the bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
for R in 1 2; do for V in new start; do
    unset RTOW_LIB_PATH; [ $V != new ] && export RTOW_LIB_PATH=$B/librtow_hip_$V.so
    python bench.py --rng per-sample --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/persample_${V}_$R.json 2> $O/persample_${V}_$R.err
    python bench.py --rng per-sample-xoroshiro --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/xoroshiro_${V}_$R.json 2> $O/xoroshiro_${V}_$R.err
    python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/mesh_${V}_$R.json 2> $O/mesh_${V}_$R.err
done; done; unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06p/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-28s %s" % (k, res[k]))
PY
timeout 1700 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/bench_mesh.json 2> $O/bench_mesh.err
python - <<'PY'
import json
for n in ("bench_driver_command", "bench_mesh"):
    d = json.loads(open("gpurun_out/r06p/%s.json" % n).read().strip().splitlines()[-1]); print(n, d["value"], {k: d[k]["value"] for k in ("plain_batches", "chain2", "plain_two_in_flight", "group_fold", "per_sample", "per_sample_xoroshiro") if k in d})
    if "host_default" in d: print("  host_default", {k: d["host_default"][k]["value"] for k in ("chain", "group_fold", "adaptive")})
PY
