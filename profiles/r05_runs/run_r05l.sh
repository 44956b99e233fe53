# round 5, GPU call l: final build (a tile's tickets most expensive first by default): the whole -m gpu suite, smoke, the driver's command in full (twice), the other configs, the headline profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -8 > $O/tests_gpu.log; cat $O/tests_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
for R in 1 2; do python bench.py --steps 20 --warmup 5 > $O/bench_driver_command_$R.json 2> $O/bench_driver_command_$R.err; done
for C in 3 4 5; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_c$C.json 2> $O/bench_c$C.err; done
python bench.py --scene mesh --steps 8 --warmup 4 --no-cpu-baseline --no-extras > $O/bench_mesh.json 2> $O/bench_mesh.err
RTOW_BENCH_DEBUG_SHARED_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 4 --warmup 1 --width 480 --height 270 --spp 16 > $O/bench_2ranks_debug.json 2> $O/bench_2ranks_debug.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05l/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d.get("value"), d.get("ms_per_step"), d.get("config", {}).get("scheduler_tune"), (d.get("config", {}).get("self_check") or {}).get("bit_identical_to_the_same_sub_batches_on_one_gpu"))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
bash profiles/collect.sh r05 10 > $O/collect_r05.log 2>&1; tail -n 2 $O/collect_r05.log
python profiles/emulate_partitions.py --config 2 > $O/partitions_c2.log 2>&1; tail -n 25 $O/partitions_c2.log
