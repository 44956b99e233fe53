# round 6, GPU call aa: lanes in a hurry as one mask per trip + their wave at priority 3 + the deep plain-launch thresholds chosen by the library (no --tune), against the build without
# any of it (nocode = the last committed kernel).  The reference host's configuration (three legs), depth 32 chains / plain launches, and what must not move: headline, groups, C4, C5.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aa; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
D=$(pwd)/raytracing-in-one-weekend_amd/csrc/build
run() {  # name lib c
  local NAME=$1 LIB=$2 R=$3
  for L in host_default_adaptive host_default_chain host_default_group group_fold plain_two_in_flight; do
    RTOW_LIB_PATH=$LIB python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/${L}_${NAME}_$R.json 2> $O/${L}_${NAME}_$R.err
  done
  RTOW_LIB_PATH=$LIB python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c2_${NAME}_$R.json 2> $O/c2_${NAME}_$R.err
  RTOW_LIB_PATH=$LIB python bench.py --steps 20 --warmup 5 --depth 32 --no-cpu-baseline --no-extras > $O/depth32_${NAME}_$R.json 2> $O/depth32_${NAME}_$R.err
  RTOW_LIB_PATH=$LIB python bench.py --steps 20 --warmup 5 --depth 32 --spp 50 --chain 1 --no-cpu-baseline --no-extras > $O/spp50depth32plain_${NAME}_$R.json 2> $O/spp50depth32plain_${NAME}_$R.err
  RTOW_LIB_PATH=$LIB python bench.py --steps 20 --warmup 5 --depth 24 --no-cpu-baseline --no-extras > $O/depth24_${NAME}_$R.json 2> $O/depth24_${NAME}_$R.err
  RTOW_LIB_PATH=$LIB python bench.py --config 5 --depth 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/c5depth32_${NAME}_$R.json 2> $O/c5depth32_${NAME}_$R.err
  RTOW_LIB_PATH=$LIB python bench.py --config 4 --depth 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/c4depth32_${NAME}_$R.json 2> $O/c4depth32_${NAME}_$R.err
  RTOW_LIB_PATH=$LIB python bench.py --scene mesh --depth 32 --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/meshdepth32_${NAME}_$R.json 2> $O/meshdepth32_${NAME}_$R.err
}
for R in 1 2 3; do
  run nocode $D/librtow_hip_nohurry.so $R
  run new "" $R
done
for C in 6 8 14; do
  for L in host_default_adaptive host_default_chain; do
    RTOW_URGENT_RAYS_PER_SAMPLE=$C python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/${L}_new-c${C}_1.json 2> $O/${L}_new-c${C}_1.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06aa/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
