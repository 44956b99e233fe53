# round 6, GPU call r: soak of the round's last build (multi-workgroup chunk order, 16-visit slices everywhere): whole frames x 2, chains, 3 000 + 1 000 heavy fuzz seeds,
# the GPU suite the way the driver runs it (serial, -x), and C3 / C4 / C5 lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu_serial.log 2>&1; tail -4 $O/pytest_gpu_serial.log
timeout 2000 python tests/soak_frames.py 2.0 > $O/soak_frames_x2.log 2>&1; tail -1 $O/soak_frames_x2.log
timeout 900 python tests/soak_chain.py 1.0 > $O/soak_chain.log 2>&1; tail -1 $O/soak_chain.log
RTOW_FUZZ_SEEDS=3000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_3000.log 2>&1; tail -1 $O/fuzz_3000.log
RTOW_FUZZ_HEAVY=1 RTOW_FUZZ_SEEDS=1000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_1000_heavy.log 2>&1; tail -1 $O/fuzz_1000_heavy.log
for C in 3 4 5; do ST=20; WU=5; [ $C = 3 ] && ST=4 && WU=2; python bench.py --config $C --steps $ST --warmup $WU --no-cpu-baseline --no-extras > $O/bench_c$C.json 2> $O/bench_c$C.err; done
python - <<'PY'
import json
for n in ("bench_c3", "bench_c4", "bench_c5"):
    d = json.loads(open("gpurun_out/r06r/%s.json" % n).read().strip().splitlines()[-1]); print(n, d["value"])
PY
