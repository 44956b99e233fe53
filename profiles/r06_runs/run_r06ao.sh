# round 6, GPU call ao: lanes in a hurry (by rate) for the moving-sphere kind too?  The moving-spheres scene (Book 2: 80 % of the spheres move, aperture 0.05) in the reference
# host's configuration, and C5 at depth 32; the build with the twins of both sphere kinds against the committed one (static spheres only: k0), same box, two rounds.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ao; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
D=$(pwd)/raytracing-in-one-weekend_amd/csrc/build
for R in 1 2; do
  for NAME in k0 k01; do
    LIB=""; [ $NAME = k0 ] && LIB=$D/librtow_hip_k0.so
    for L in host_default_adaptive host_default_chain; do
      RTOW_LIB_PATH=$LIB python bench.py --scene moving --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/moving_${L}_${NAME}_$R.json 2> $O/moving_${L}_${NAME}_$R.err
    done
    RTOW_LIB_PATH=$LIB python bench.py --scene moving --depth 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/moving_depth32_${NAME}_$R.json 2> $O/moving_depth32_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --config 5 --depth 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/c5depth32_${NAME}_$R.json 2> $O/c5depth32_${NAME}_$R.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06ao/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f)[:-5].rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
