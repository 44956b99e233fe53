# round 6, GPU call f: whole GPU suite on the build with the pixel-boundary company (generic / per-sample variants only, K = 4 by rule) and the pinhole twins beyond LDS;
# then per-sample units of 16 (the policy) against a timing build with units of 64; host-default legs and C5 / C2 against the round's starting build.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
timeout 1500 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
for R in 1 2; do for V in new start sg64; do
  unset RTOW_LIB_PATH; [ $V != new ] && export RTOW_LIB_PATH=$B/librtow_hip_$V.so
  python bench.py --rng per-sample --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/persample_${V}_$R.json 2> $O/persample_${V}_$R.err
  python bench.py --rng per-sample-xoroshiro --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/xoroshiro_${V}_$R.json 2> $O/xoroshiro_${V}_$R.err
  if [ $V != sg64 ]; then
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
    python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c5_${V}_$R.json 2> $O/c5_${V}_$R.err
    for L in host_default_group host_default_chain host_default_adaptive; do
      python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/${L}_${V}_$R.json 2> $O/${L}_${V}_$R.err
    done
  fi
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06f/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-36s %s" % (k, res[k]))
PY
