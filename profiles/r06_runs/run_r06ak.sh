# round 6, GPU call ak: the driver's command sixteen times in a row on one box, the round's last build (headline kernels unchanged by the lanes in a hurry: the spread is the box's)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ak; mkdir -p $O
for I in $(seq 1 16); do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/run_$I.json 2> $O/run_$I.err; done
python - <<'PY'
import json, glob
v = []
for i in range(1, 17):
    v.append(json.loads(open("gpurun_out/r06ak/run_%d.json" % i).read().strip().splitlines()[-1])["value"])
out = {"command": "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras, sixteen times in a row on one box (the round's last build)",
       "msamples_per_s": v, "mean": sum(v) / len(v), "min": min(v), "max": max(v), "within_half_a_percent_of_the_median": sum(1 for x in v if abs(x / sorted(v)[len(v) // 2] - 1) <= 0.005)}
json.dump(out, open("gpurun_out/r06ak/repeatability.json", "w")); print(out)
PY
