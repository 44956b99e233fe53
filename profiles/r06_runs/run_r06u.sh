# round 6, GPU call u: r06t's one surprise - the adaptive {1, 50} schedule of the reference host's configuration ran 7 % faster with the walk handing over at 4 candidates instead of 3
# (one run).  Repeat it, with neighbours and combinations, three rounds, baseline interleaved; the chain and group legs beside it.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
for R in 1 2 3; do
  for T in "24,32,1,32,28,1,3,3,16" "24,32,1,32,28,1,4,3,16" "24,32,1,32,28,1,5,3,16" "24,32,1,32,28,1,6,3,16" "24,32,1,24,28,1,4,3,16" "24,32,1,24,28,1,4,3,12" "16,32,1,24,28,1,4,3,16" "24,32,1,24,28,1,3,3,16" "24,32,1,32,28,1,4,3,12"; do
    N=$(echo $T | tr ',' '_')
    for L in host_default_adaptive host_default_chain host_default_group; do
      python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline --tune $T > $O/${L}_${N}_$R.json 2> $O/${L}_${N}_$R.err
    done
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06u/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-64s %s" % (k, res[k]))
PY
