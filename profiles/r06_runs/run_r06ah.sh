# round 6, GPU call ah: lanes in a hurry by rate, second sweep (r06ag: the larger c and the smaller k the better, up to c = 14, k = 2)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ah; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
D=$(pwd)/raytracing-in-one-weekend_amd/csrc/build
for R in 1 2; do
  for NAME in shipped rate_c14_k2 rate_c14_k1 rate_c18_k2 rate_c18_k1 rate_c24_k2; do
    LIB=""; [ $NAME != shipped ] && LIB=$D/librtow_hip_$NAME.so
    for L in host_default_adaptive host_default_chain; do
      RTOW_LIB_PATH=$LIB python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/${L}_${NAME}_$R.json 2> $O/${L}_${NAME}_$R.err
    done
    RTOW_LIB_PATH=$LIB python bench.py --depth 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/depth32_${NAME}_$R.json 2> $O/depth32_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --depth 32 --spp 50 --chain 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/spp50depth32plain_${NAME}_$R.json 2> $O/spp50depth32plain_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --config 4 --depth 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/c4depth32_${NAME}_$R.json 2> $O/c4depth32_${NAME}_$R.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06ah/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f)[:-5].rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
