# round 6, GPU call m: the build with history rows in HBM where LDS is short (the heavy fuzz of call l found the launches the first plan refused): whole GPU suite, heavy and
# plain fuzz, frame and chain soak, the bench lines again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
RTOW_FUZZ_HEAVY=1 RTOW_FUZZ_SEEDS=2000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_2000_heavy.log 2>&1; tail -1 $O/fuzz_2000_heavy.log
RTOW_FUZZ_SEEDS=3000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_3000.log 2>&1; tail -1 $O/fuzz_3000.log
timeout 1500 python tests/soak_frames.py 1.0 > $O/soak_frames.log 2>&1; tail -2 $O/soak_frames.log
timeout 900 python tests/soak_chain.py 1.0 > $O/soak_chain.log 2>&1; tail -1 $O/soak_chain.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
for C in 3 4 5; do ST=20; WU=5; [ $C = 3 ] && ST=4 && WU=2; python bench.py --config $C --steps $ST --warmup $WU --no-cpu-baseline --no-extras > $O/bench_c$C.json 2> $O/bench_c$C.err; done
python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/bench_mesh.json 2> $O/bench_mesh.err
python bench.py --scene mesh --depth 32 --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/bench_mesh_depth32.json 2> $O/bench_mesh_depth32.err
python - <<'PY'
import json
for n in ("bench_driver_command", "bench_c3", "bench_c4", "bench_c5", "bench_mesh", "bench_mesh_depth32"):
    try:
        d = json.loads(open("gpurun_out/r06m/%s.json" % n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], {k: d[k]["value"] for k in ("plain_batches", "chain2", "plain_two_in_flight", "group_fold", "per_sample", "per_sample_xoroshiro") if k in d})
        if "host_default" in d: print("  host_default", {k: d["host_default"][k]["value"] for k in ("chain", "group_fold", "adaptive")})
    except Exception as e: print(n, "FAILED", e)
PY
