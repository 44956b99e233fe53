# round 6, GPU call b: per-launch LDS plan (a stack row per tree level, path history beyond depth 8 in LDS rows, reference-count walk in its own variant) + the round's mesh changes
# (prefetch off).  Whole GPU suite on the new build, then new against the round's starting build, same box, alternating: headline, host-default legs, C4, C5, mesh.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
timeout 1500 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
for R in 1 2; do for V in new start; do
  unset RTOW_LIB_PATH; [ $V = start ] && export RTOW_LIB_PATH=$B/librtow_hip_start.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  for L in host_default_group host_default_chain host_default_adaptive group_fold; do
    python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/${L}_${V}_$R.json 2> $O/${L}_${V}_$R.err
  done
  for C in 4 5; do python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c${C}_${V}_$R.json 2> $O/c${C}_${V}_$R.err; done
  python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/mesh_${V}_$R.json 2> $O/mesh_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06b/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-36s %s" % (k, res[k]))
PY
