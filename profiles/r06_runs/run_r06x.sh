# round 6, GPU call x: lanes in a hurry - a pixel that has taken more than c x (samples per batch) rays stops waiting for company at the stage thresholds.  The reference host's
# configuration as adaptive / chained / grouped launches; c from the environment (experiment build), 0 = off; the build without the code beside it; two threshold sets.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
A="24,32,1,32,28,1,3,3,16"; B="16,32,1,24,28,1,4,3,16"
NOH=$(pwd)/raytracing-in-one-weekend_amd/csrc/build/librtow_hip_nohurry.so
for R in 1 2; do
  for L in host_default_adaptive host_default_chain host_default_group; do
    RTOW_LIB_PATH=$NOH python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline --tune $A > $O/${L}_nocode_A_$R.json 2> $O/${L}_nocode_A_$R.err
    for C in 0 3 4 6 8 12; do
      for NAME in A B; do
        T=${!NAME}
        RTOW_URGENT_RAYS_PER_SAMPLE=$C python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline --tune $T > $O/${L}_c${C}_${NAME}_$R.json 2> $O/${L}_c${C}_${NAME}_$R.err
      done
    done
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06x/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append((round(d["value"], 1), d["ms_per_step"]))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -3
