# round 6, GPU call v: r06u - the reference host's configuration as chains / adaptive launches wants HIT from 24/64 of the live lanes and the walk's hand-over at 4 candidates
# (+7 ... +9 %), as batch groups it does not (-2.8 %).  What decides: the samples per batch (50 against 256), the trace depth (32: kernel <...,32,...>), the 16-byte records?
# Main measurement (chains of 10) with --spp / --depth moved one at a time, two settings, three rounds; plain launches (chain 1) beside it.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
A="24,32,1,32,28,1,3,3,16"; B="16,32,1,24,28,1,4,3,16"; C="24,32,1,32,28,1,4,3,16"; D="24,32,1,24,28,1,3,3,16"
for R in 1 2 3; do
  for NAME in A B C D; do
    T=${!NAME}
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $T > $O/c2_${NAME}_$R.json 2> $O/c2_${NAME}_$R.err
    python bench.py --steps 20 --warmup 5 --spp 50 --no-cpu-baseline --no-extras --tune $T > $O/spp50_${NAME}_$R.json 2> $O/spp50_${NAME}_$R.err
    python bench.py --steps 20 --warmup 5 --depth 32 --no-cpu-baseline --no-extras --tune $T > $O/depth32_${NAME}_$R.json 2> $O/depth32_${NAME}_$R.err
    python bench.py --steps 20 --warmup 5 --spp 50 --depth 32 --no-cpu-baseline --no-extras --tune $T > $O/spp50depth32_${NAME}_$R.json 2> $O/spp50depth32_${NAME}_$R.err
    python bench.py --steps 20 --warmup 5 --spp 50 --depth 32 --chain 1 --no-cpu-baseline --no-extras --tune $T > $O/spp50depth32plain_${NAME}_$R.json 2> $O/spp50depth32plain_${NAME}_$R.err
    python bench.py --steps 10 --warmup 3 --chain 1 --no-cpu-baseline --no-extras --tune $T > $O/c2plain_${NAME}_$R.json 2> $O/c2plain_${NAME}_$R.err
    python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-extras --tune $T > $O/c3_${NAME}_$R.json 2> $O/c3_${NAME}_$R.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06v/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-40s %s" % (k, res[k]))
PY
