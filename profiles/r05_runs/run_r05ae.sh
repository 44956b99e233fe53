# round 5, GPU call ae: plain launches with slot tickets (development knob: K slots per pull from h sixteenths of the order on, one at a time before it and over the last eighth) against one chunk per pull, and the build before (prev)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ae; mkdir -p $O
python -m pytest tests/test_gpu_group.py tests/test_gpu_chain.py tests/test_gpu_golden.py -q -x -n 4 2>&1 | tail -2
PREV=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_prev.so
for R in 1 2; do
  for V in new prev; do unset RTOW_LIB_PATH; [ $V = prev ] && export RTOW_LIB_PATH=$PREV
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
    python bench.py --steps 10 --warmup 3 --chain 1 --no-cpu-baseline --no-extras > $O/plain_${V}_$R.json 2> $O/plain_${V}_$R.err
  done
  unset RTOW_LIB_PATH
  for KH in "2 4" "2 2" "4 4" "2 6" "3 4"; do set -- $KH; T=0,0,0,0,0,0,0,$((3 + 256 * $1 + 4096 * $2)),0
    python bench.py --steps 10 --warmup 3 --chain 1 --no-cpu-baseline --no-extras --tune $T > $O/plain_K$1h$2_$R.json 2> $O/plain_K$1h$2_$R.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05ae/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d.get("value"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-300:])
for k in sorted(res): print("%-30s %s" % (k, res[k]))
PY
