# round 6, GPU call e: pinhole twins of the sphere kernels (default build) against the same build without them (nopin); relaxed trips (every 16th / 64th trip runs every
# stage from one lane on; A/B build, not shipped); ALL_LAMBERT timing build (ceiling of any regrouping of the HIT classes); then 12 consecutive driver commands (dips).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
B=raytracing-in-one-weekend_amd/csrc/build
T="24,32,1,32,28,1,3"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
for R in 1 2 3; do for V in new nopin relax16 relax64 alllambert; do
  unset RTOW_LIB_PATH; X=""
  case $V in new) ;; nopin) export RTOW_LIB_PATH=$B/librtow_hip_nopin.so ;; alllambert) export RTOW_LIB_PATH=$B/librtow_hip_alllambert.so ;;
    relax16) export RTOW_LIB_PATH=$B/librtow_hip_relax.so; X="--tune $T,$((3 + 4096 + 65536 * 4)),16" ;;
    relax64) export RTOW_LIB_PATH=$B/librtow_hip_relax.so; X="--tune $T,$((3 + 4096 + 65536 * 6)),16" ;; esac
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $X > $O/c2_${V}_$R.json 2> $O/c2_${V}_$R.err
  python bench.py --steps 10 --warmup 3 --chain 1 --no-cpu-baseline --no-extras $X > $O/c2plain_${V}_$R.json 2> $O/c2plain_${V}_$R.err
  python bench.py --only-leg group_fold --chain 10 --steps 20 --no-cpu-baseline $X > $O/groupfold_${V}_$R.json 2> $O/groupfold_${V}_$R.err
  if [ $V = new -o $V = nopin ]; then
    python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/c4_${V}_$R.json 2> $O/c4_${V}_$R.err
    python bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/c3_${V}_$R.json 2> $O/c3_${V}_$R.err
  fi
done; done
unset RTOW_LIB_PATH
for I in 1 2 3 4 5 6 7 8 9 10 11 12; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/rep_new_$I.json 2> $O/rep_new_$I.err; done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06e/*_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-36s %s" % (k, res[k]))
PY
