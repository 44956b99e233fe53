# round 6, GPU call o: (1) walk slice of the mesh kernel below 24 visits; (2) the chunk order built by several workgroups where a launch has more than 65 536 chunks
# (per-sample policies at 1080p: 518 400): per-sample bench against the round's starting build, and the per-sample / order tests.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_matrix.py tests/test_gpu_regroup.py -q -n 4 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
MESH="--scene mesh --steps 8 --warmup 4 --chain 4 --no-cpu-baseline --no-extras"
python bench.py $MESH > /dev/null 2>&1
for R in 1 2; do
  for S in 8 12 16 20 24; do python bench.py $MESH --tune 8,48,1,1,8,1,3,3,$S > $O/mesh_s${S}_$R.json 2> $O/mesh_s${S}_$R.err; done
  for V in new start; do
    unset RTOW_LIB_PATH; [ $V != new ] && export RTOW_LIB_PATH=$B/librtow_hip_$V.so
    python bench.py --rng per-sample --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/persample_${V}_$R.json 2> $O/persample_${V}_$R.err
    python bench.py --rng per-sample-xoroshiro --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/xoroshiro_${V}_$R.json 2> $O/xoroshiro_${V}_$R.err
  done; unset RTOW_LIB_PATH
  for S in 12 16 20; do python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune 24,32,1,32,28,1,3,3,$S > $O/c4_s${S}_$R.json 2> $O/c4_s${S}_$R.err; done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06o/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-28s %s" % (k, res[k]))
PY
