# round 6, GPU call n: the mesh kernel is the rank-rule kernel now (no scratch, 116 VGPRs): are round 3's walk slice (24 visits), hand-over (3 candidates) and walk threshold
# (3/4 of the live lanes) still its best?  One library, --tune, same box, two rounds.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
MESH="--scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras"
python bench.py $MESH > /dev/null 2>&1
for R in 1 2; do
  for T in "8,48,1,1,8,1,3,3,24" "8,48,1,1,8,1,3,3,16" "8,48,1,1,8,1,3,3,32" "8,48,1,1,8,1,3,3,48" "8,48,1,1,8,1,2,3,24" "8,48,1,1,8,1,4,3,24" "8,48,1,1,8,1,5,3,32" "8,32,1,1,8,1,3,3,24" "8,56,1,1,8,1,3,3,24" "8,40,1,16,8,1,3,3,24" "16,48,1,1,16,1,3,3,24" "1,48,1,1,1,1,3,3,24"; do
    N=$(echo $T | tr ',' '_')
    python bench.py $MESH --tune $T > $O/mesh_${N}_$R.json 2> $O/mesh_${N}_$R.err
  done
  python bench.py $MESH > $O/mesh_default_$R.json 2> $O/mesh_default_$R.err
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06n/mesh_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f)[5:].rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res, key=lambda k: -sum(res[k])): print("%-28s %s" % (k, res[k]))
PY
