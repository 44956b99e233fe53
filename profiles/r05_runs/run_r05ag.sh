# round 5, GPU call ag: the round's last build (whole material record in HIT beyond LDS) through the whole -m gpu suite and smoke; bench lines of every config; counters of C4 and the mesh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ag; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -4 > $O/tests_gpu.log; cat $O/tests_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
for C in 3 4 5; do ST=20; WU=5; [ $C = 3 ] && ST=4 && WU=2; python bench.py --config $C --steps $ST --warmup $WU --no-cpu-baseline --no-extras > $O/bench_c$C.json 2> $O/bench_c$C.err; done
python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/bench_mesh.json 2> $O/bench_mesh.err
python - <<'PY'
import json
for n in ("bench_driver_command", "bench_c3", "bench_c4", "bench_c5", "bench_mesh"):
    try:
        d = json.loads(open("gpurun_out/r05ag/%s.json" % n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], {k: d[k]["value"] for k in ("plain_batches", "chain2", "group_fold") if k in d})
    except Exception as e: print(n, "FAILED", e)
PY
POST=0 L2=1 bash profiles/collect.sh r05_c4 10 --config 4 > $O/collect_c4.log 2>&1
POST=0 L2=1 bash profiles/collect.sh r05_mesh 4 --scene mesh > $O/collect_mesh.log 2>&1
for f in $O/collect_*.log; do echo $f; tail -n 2 $f; done
