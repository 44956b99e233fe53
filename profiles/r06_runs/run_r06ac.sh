# round 6, GPU call ac: lanes in a hurry as twins of the generic variants (GEO bit 4: plain / chained launches; groups keep the variants without the code).  (1) the reference host's
# configuration, new against the last committed kernel (nocode);  (2) moving spheres / 10 000 spheres at depth 32: hurry and the deep plain-launch thresholds apart
# (RTOW_URGENT_RAYS_PER_SAMPLE=0: no bound; --tune with the built-in values: the library's threshold rule off);  (3) the new tests.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ac; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
D=$(pwd)/raytracing-in-one-weekend_amd/csrc/build
A="24,32,1,32,28,1,3,3,16"
for R in 1 2 3; do
  for NAME in nocode new; do
    LIB=""; [ $NAME = nocode ] && LIB=$D/librtow_hip_nohurry.so
    for L in host_default_adaptive host_default_chain host_default_group; do
      RTOW_LIB_PATH=$LIB python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/${L}_${NAME}_$R.json 2> $O/${L}_${NAME}_$R.err
    done
  done
  for CFG in 5 4; do
    X="--config $CFG --depth 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
    RTOW_LIB_PATH=$D/librtow_hip_nohurry.so python bench.py $X > $O/c${CFG}depth32_nocode_$R.json 2> $O/c${CFG}depth32_nocode_$R.err
    RTOW_URGENT_RAYS_PER_SAMPLE=0 python bench.py $X --tune $A > $O/c${CFG}depth32_neither_$R.json 2> $O/c${CFG}depth32_neither_$R.err
    RTOW_URGENT_RAYS_PER_SAMPLE=0 python bench.py $X > $O/c${CFG}depth32_thresholds_$R.json 2> $O/c${CFG}depth32_thresholds_$R.err
    python bench.py $X --tune $A > $O/c${CFG}depth32_hurry_$R.json 2> $O/c${CFG}depth32_hurry_$R.err
    python bench.py $X > $O/c${CFG}depth32_both_$R.json 2> $O/c${CFG}depth32_both_$R.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06ac/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_variants.py tests/test_gpu_group.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -4
