# round 5, GPU call r: compiler-flag variants of the whole library against the shipped build, driver's command + C5, same box, two alternating rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
for R in 1 2; do for V in head memclause iterilp itermaxocc nounroll nopostsched; do
  unset RTOW_LIB_PATH; [ $V != head ] && export RTOW_LIB_PATH=$B/librtow_hip_$V.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c5_${V}_$R.json 2> $O/c5_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05r/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
