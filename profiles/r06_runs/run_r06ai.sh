# round 6, GPU call ai: the round's last build - lanes in a hurry by rate (18 rays per sample done, 14 with the tree beyond LDS).  Same-box figures against the kernel before
# (nocode), the driver's command (with a kernel trace of it), smoke, the GPU suite as the driver runs it, and the soak: whole frames x 2, chains, 3 000 + 1 000 heavy fuzz seeds.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ai; mkdir -p $O
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
    RTOW_LIB_PATH=$LIB python bench.py --depth 32 --spp 50 --chain 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/spp50depth32plain_${NAME}_$R.json 2> $O/spp50depth32plain_${NAME}_$R.err
  done
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06ai/*_[123].json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f)[:-5].rsplit("_", 1)[0]].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06ai/bench_driver_command.json").read().strip().splitlines()[-1])
print("driver", d["value"], "per_sample", d.get("per_sample", {}).get("value"), "host_default", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("host_default", {}).items() if k in ("chain", "group_fold", "adaptive")})
PY
REPO=$(pwd); (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace -o bench -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $REPO/$O/trace.log 2>&1)
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/driver_command_kernel_stats.csv \;
rm -rf $O/trace
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 1800 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu_serial.log 2>&1; tail -5 $O/pytest_gpu_serial.log
timeout 2000 python tests/soak_frames.py 2.0 > $O/soak_frames_x2.log 2>&1; tail -1 $O/soak_frames_x2.log
timeout 1200 python tests/soak_chain.py 1.0 > $O/soak_chain.log 2>&1; tail -1 $O/soak_chain.log
RTOW_FUZZ_SEEDS=3000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_3000.log 2>&1; tail -1 $O/fuzz_3000.log
RTOW_FUZZ_HEAVY=1 RTOW_FUZZ_SEEDS=1000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 4 > $O/fuzz_1000_heavy.log 2>&1; tail -1 $O/fuzz_1000_heavy.log
