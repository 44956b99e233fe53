# round 5, GPU call aj: the round's last build against the build this session started from (commit 82bc725), same box, three alternating rounds, every workload of the bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05aj; mkdir -p $O
START=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_start.so
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
for R in 1 2 3; do for V in last start; do
  unset RTOW_LIB_PATH; [ $V = start ] && export RTOW_LIB_PATH=$START
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  for C in 4 5; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c${C}_${V}_$R.json 2> $O/c${C}_${V}_$R.err; done
  python bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/c3_${V}_$R.json 2> $O/c3_${V}_$R.err
  python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/mesh_${V}_$R.json 2> $O/mesh_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05aj/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d["value"])
        for k in ("plain_batches", "chain2", "group_fold"):
            if k in d: res[name + " " + k].append(d[k]["value"])
        if "host_default" in d:
            for k in ("chain", "group_fold", "adaptive"):
                if k in d["host_default"]: res[name + " host_default." + k].append(d["host_default"][k]["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-40s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
