# round 6, GPU call w: is the reference host's configuration as plain / chained launches bound by its slowest pixel (50 samples of up to 32 segments in a row on one lane)?
# (1) a quarter of the pixels: does a step take a quarter of the time?  (2) thresholds that favour latency over occupancy.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06w; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
for R in 1 2; do
  for WH in "1920 1080" "960 540" "480 270"; do
    set -- $WH
    for L in host_default_adaptive host_default_chain host_default_group; do
      python bench.py --only-leg $L --chain 10 --steps 20 --width $1 --height $2 --no-cpu-baseline > $O/${L}_w$1_$R.json 2> $O/${L}_w$1_$R.err
    done
  done
  for T in "1,1,1,1,1,1,3,3,16" "8,16,1,8,8,1,4,3,16" "16,32,1,24,28,1,4,3,32" "16,16,1,16,16,1,4,3,16" "16,32,1,24,28,1,4,3,16" "16,32,1,24,28,1,4,259,16" "16,32,1,24,28,1,4,4099,16"; do
    N=$(echo $T | tr ',' '_')
    for L in host_default_adaptive host_default_chain; do
      python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline --tune $T > $O/${L}_${N}_$R.json 2> $O/${L}_${N}_$R.err
    done
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06w/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append((round(d["value"], 1), d["ms_per_step"]))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-64s %s" % (k, res[k]))
PY
