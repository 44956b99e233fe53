# round 6, GPU call a: the triangle-mesh kernel - compact hot / cold triangle records in leaf order, prefetch of pushed nodes / listed triangles, tie watch for all-triangle scenes
# (rank-rule kernels + fix-up launch instead of the exact-tie kernels for every pixel).  Whole GPU suite on the new build, then the mesh bench per build variant, same box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
timeout 1500 python -m pytest tests -m gpu -x -q -n 4 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
MESH="--scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras"
python bench.py $MESH > /dev/null 2>&1      # the box's first run
for R in 1 2; do for V in new start pf0 pf1 pf2 cold0 h48 newalways; do
  unset RTOW_LIB_PATH; X=""
  case $V in new) ;; newalways) X="--context-flags 1" ;; *) export RTOW_LIB_PATH=$B/librtow_hip_$V.so ;; esac
  timeout 600 python bench.py $MESH $X > $O/mesh_${V}_$R.json 2> $O/mesh_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06a/mesh_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-24s %s" % (k, res[k]))
PY
