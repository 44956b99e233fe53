# round 6, GPU call s: nodes 80 bytes apart in LDS (a ds_read_b128 of 16 lanes then spreads over all 64 banks instead of 16) against the same build with the blob's 64-byte
# stride (n64), same box, three alternating rounds; parity subset first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_variants.py -q -n 4 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
for R in 1 2 3; do for V in new n64; do
  unset RTOW_LIB_PATH; [ $V != new ] && export RTOW_LIB_PATH=$B/librtow_hip_$V.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  python bench.py --steps 10 --warmup 3 --chain 1 --no-cpu-baseline --no-extras > $O/c2plain_${V}_$R.json 2> $O/c2plain_${V}_$R.err
  python bench.py --only-leg group_fold --chain 10 --steps 20 --no-cpu-baseline > $O/groupfold_${V}_$R.json 2> $O/groupfold_${V}_$R.err
  python bench.py --only-leg host_default_group --chain 10 --steps 20 --no-cpu-baseline > $O/hostdefault_${V}_$R.json 2> $O/hostdefault_${V}_$R.err
  python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c5_${V}_$R.json 2> $O/c5_${V}_$R.err
  python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c4_${V}_$R.json 2> $O/c4_${V}_$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06s/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-24s %s" % (k, res[k]))
PY
