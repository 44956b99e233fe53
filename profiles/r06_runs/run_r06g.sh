# round 6, GPU call g: packed 16 + 2-bit traversal stack for all-triangle scenes up to 262 144 nodes (twice the tree top in LDS): whole GPU suite (incl. the 288 002-triangle
# grid that keeps 32-bit rows), then the mesh bench: packed (new) against the same build with 32-bit rows (nopack) and the round's starting build, same box, three alternating rounds.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
timeout 1700 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
MESH="--scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras"
python bench.py $MESH > /dev/null 2>&1
for R in 1 2 3; do for V in new nopack start; do
  unset RTOW_LIB_PATH; [ $V != new ] && export RTOW_LIB_PATH=$B/librtow_hip_$V.so
  python bench.py $MESH > $O/mesh_${V}_$R.json 2> $O/mesh_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list); tset = {}
for f in sorted(glob.glob("gpurun_out/r06g/mesh_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1)); tset.setdefault(name, []).append(d["config"]["threshold_set"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-24s %s thresholds %s" % (k, res[k], tset[k]))
PY
