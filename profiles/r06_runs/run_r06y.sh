# round 6, GPU call y: lanes in a hurry, second form - while a wave holds such a lane, the stages no hurried lane waits in need more company (RTOW_URGENT_RAISE / 64 of the live lanes),
# so that its trips hold little but what the hurried lane needs.  c = 8 ... 24 rays per sample; experiment build, knobs from the environment.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
A="24,32,1,32,28,1,3,3,16"; B="16,32,1,24,28,1,4,3,16"
for R in 1 2; do
  for L in host_default_adaptive host_default_chain; do
    for C in 0 8 12 16 24; do
      for RAISE in 0 32 48 64; do
        if [ $C = 0 ] && [ $RAISE != 0 ]; then continue; fi
        for NAME in A B; do
          T=${!NAME}
          RTOW_URGENT_RAISE=$RAISE RTOW_URGENT_RAYS_PER_SAMPLE=$C python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline --tune $T > $O/${L}_c${C}_r${RAISE}_${NAME}_$R.json 2> $O/${L}_c${C}_r${RAISE}_${NAME}_$R.err
        done
      done
    done
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06y/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append((round(d["value"], 1), d["ms_per_step"]))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
