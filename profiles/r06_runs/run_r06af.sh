# round 6, GPU call af: soak of the build with the lanes in a hurry: whole frames x 2 (cover at depth 32 with both record formats, 6 000 spheres at depth 24 among them), chains
# (two new cases through the twins), 3 000 + 1 000 heavy fuzz seeds, and the C3 / C4 / C5 / mesh lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06af; mkdir -p $O
timeout 2000 python tests/soak_frames.py 2.0 > $O/soak_frames_x2.log 2>&1; tail -1 $O/soak_frames_x2.log
timeout 1200 python tests/soak_chain.py 1.0 > $O/soak_chain.log 2>&1; tail -1 $O/soak_chain.log
RTOW_FUZZ_SEEDS=3000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_3000.log 2>&1; tail -1 $O/fuzz_3000.log
RTOW_FUZZ_HEAVY=1 RTOW_FUZZ_SEEDS=1000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_1000_heavy.log 2>&1; tail -1 $O/fuzz_1000_heavy.log
for C in 3 4 5; do ST=20; WU=5; [ $C = 3 ] && ST=4 && WU=2; python bench.py --config $C --steps $ST --warmup $WU --no-cpu-baseline --no-extras > $O/bench_c$C.json 2> $O/bench_c$C.err; done
python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/bench_mesh.json 2> $O/bench_mesh.err
python bench.py --scene mesh --depth 32 --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/bench_mesh_depth32.json 2> $O/bench_mesh_depth32.err
python - <<'PY'
import json
for n in ("bench_c3", "bench_c4", "bench_c5", "bench_mesh", "bench_mesh_depth32"):
    d = json.loads(open("gpurun_out/r06af/%s.json" % n).read().strip().splitlines()[-1]); print(n, d["value"])
PY
