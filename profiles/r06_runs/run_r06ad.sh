# round 6, GPU call ad: validation of the build with the lanes in a hurry (twins of the static-sphere generic variants) and the deep plain-launch thresholds:
# final figures against the last committed kernel (nocode) on one box, a kernel trace of the reference host's configuration as chains, the driver's command, smoke, the GPU suite.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ad; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
D=$(pwd)/raytracing-in-one-weekend_amd/csrc/build
for R in 1 2 3; do
  for NAME in nocode new; do
    LIB=""; [ $NAME = nocode ] && LIB=$D/librtow_hip_nohurry.so
    for L in host_default_adaptive host_default_chain host_default_group; do
      RTOW_LIB_PATH=$LIB python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/${L}_${NAME}_$R.json 2> $O/${L}_${NAME}_$R.err
    done
    RTOW_LIB_PATH=$LIB python bench.py --config 5 --depth 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/c5depth32_${NAME}_$R.json 2> $O/c5depth32_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --config 4 --depth 32 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/c4depth32_${NAME}_$R.json 2> $O/c4depth32_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --depth 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/depth32_${NAME}_$R.json 2> $O/depth32_${NAME}_$R.err
    RTOW_LIB_PATH=$LIB python bench.py --depth 24 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/depth24_${NAME}_$R.json 2> $O/depth24_${NAME}_$R.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06ad/*_[123].json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
REPO=$(pwd); (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_chain -o bench -- python $REPO/bench.py --only-leg host_default_chain --chain 10 --steps 20 --no-cpu-baseline > $REPO/$O/trace_chain.log 2>&1)
find $O/trace_chain -name "*kernel_stats.csv" -exec cp {} $O/host_default_chain_kernel_stats.csv \;
rm -rf $O/trace_chain
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -8
