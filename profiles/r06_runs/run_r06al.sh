# round 6, GPU call al: while a wave of a plain launch holds a lane in a hurry, its other lanes take no new pixel (the wave drains, its trips hold little but what the hurried lane
# needs).  Experiment build (RTOW_HURRY_DRAIN) against the shipped one, same box, three rounds.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06al; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
D=$(pwd)/raytracing-in-one-weekend_amd/csrc/build
for R in 1 2 3; do
  for NAME in shipped drain; do
    LIB=""; [ $NAME != shipped ] && LIB=$D/librtow_hip_$NAME.so
    RTOW_LIB_PATH=$LIB python bench.py --only-leg host_default_adaptive --chain 10 --steps 20 --no-cpu-baseline > $O/host_default_adaptive_${NAME}_$R.json 2> $O/host_default_adaptive_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --depth 32 --spp 50 --chain 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/spp50depth32plain_${NAME}_$R.json 2> $O/spp50depth32plain_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --depth 32 --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/depth32plain_${NAME}_$R.json 2> $O/depth32plain_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --config 4 --depth 32 --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/c4depth32plain_${NAME}_$R.json 2> $O/c4depth32plain_${NAME}_$R.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06al/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f)[:-5].rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
