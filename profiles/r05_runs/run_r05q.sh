# round 5, GPU call q: node loads of the trees that do not fit LDS as ds_read / global_load behind a wave-uniform branch (RTOW_SPLIT_NODE_LOADS=1) against one flat load per quad: parity, then same-box A/B on C4 and the mesh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x -n 4 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
FLAT=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_flat.so
for R in 1 2 3; do for V in split flat; do
  unset RTOW_LIB_PATH; [ $V = flat ] && export RTOW_LIB_PATH=$FLAT
  python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c4_${V}_$R.json 2> $O/c4_${V}_$R.err
  python bench.py --scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras > $O/mesh_${V}_$R.json 2> $O/mesh_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05q/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
