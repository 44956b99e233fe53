# round 6, GPU call l: long soak of the final build (whole frames x 3, chains x 2, 1 000 heavy fuzz seeds, 6 000 plain ones) and eight consecutive driver commands
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
timeout 2400 python tests/soak_frames.py 3.0 > $O/soak_frames_x3.log 2>&1; tail -2 $O/soak_frames_x3.log
timeout 1200 python tests/soak_chain.py 2.0 > $O/soak_chain_x2.log 2>&1; tail -1 $O/soak_chain_x2.log
RTOW_FUZZ_SEEDS=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_6000.log 2>&1; tail -1 $O/fuzz_6000.log
RTOW_FUZZ_HEAVY=1 RTOW_FUZZ_SEEDS=1000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_1000_heavy.log 2>&1; tail -1 $O/fuzz_1000_heavy.log
for I in 1 2 3 4 5 6 7 8; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/rep_$I.json 2> $O/rep_$I.err; done
python - <<'PY'
import json
v = [json.loads(open("gpurun_out/r06l/rep_%d.json" % i).read().strip().splitlines()[-1])["value"] for i in range(1, 9)]
print(v, sum(v) / len(v), min(v), max(v))
json.dump({"command": "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras, eight times in a row on one box (the round's final build)", "msamples_per_s": v,
           "mean": sum(v) / len(v), "min": min(v), "max": max(v)}, open("gpurun_out/r06l/repeatability.json", "w"), indent=1)
PY
