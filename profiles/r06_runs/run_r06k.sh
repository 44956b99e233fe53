# round 6, GPU call k: queue slots per pull by the slots a wave has to work through (sliced group launches regressed with round 5's fixed four): the partition emulation again,
# group tests, and the whole-frame group leg (must not move)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_comm.py -q -n 4 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
for R in 1 2; do python bench.py --only-leg group_fold --chain 10 --steps 20 --no-cpu-baseline > $O/groupfold_$R.json 2> $O/groupfold_$R.err; python bench.py --only-leg host_default_group --chain 10 --steps 20 --no-cpu-baseline > $O/hostdefault_$R.json 2> $O/hostdefault_$R.err; done
timeout 900 python profiles/emulate_partitions.py --config 2 > $O/partitions_c2.json 2> $O/partitions_c2.err
python - <<'PY'
import json
for R in (1, 2):
    for n in ("groupfold", "hostdefault"):
        print(n, R, json.loads(open("gpurun_out/r06k/%s_%d.json" % (n, R)).read().strip().splitlines()[-1])["value"])
d = json.load(open("gpurun_out/r06k/partitions_c2.json"))
for key, v in d["partitions"].items(): print(key, v.get("slowest_render_ms"), v.get("group_render_ms_per_step"), v.get("predicted_speedup_vs_n1_bench"))
print(d["best_partition_per_world"])
PY
