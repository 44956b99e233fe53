# round 5, GPU call d: the whole -m gpu suite on the current build, the 16-word history variants A/B (host-default groups, adaptive schedule, depth-32 launches),
# the driver's command in full, and the headline profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -15 > $O/tests_gpu.log; cat $O/tests_gpu.log
OFF=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_h16off.so
for R in 1 2; do for L in on off; do
  if [ $L = off ]; then export RTOW_LIB_PATH=$OFF; else unset RTOW_LIB_PATH; fi
  python bench.py --only-leg host_default_group --steps 20 --chain 10 > $O/h16_${L}_group_rep$R.json 2> $O/h16_${L}_group_rep$R.err
  python bench.py --only-leg host_default_chain --steps 20 --chain 10 > $O/h16_${L}_chain_rep$R.json 2>> $O/h16_${L}_group_rep$R.err
  python bench.py --only-leg host_default_adaptive --steps 20 > $O/h16_${L}_adaptive_rep$R.json 2>> $O/h16_${L}_group_rep$R.err
  python bench.py --depth 32 --steps 10 --warmup 2 --chain 1 --no-cpu-baseline --no-extras > $O/h16_${L}_depth32plain_rep$R.json 2>> $O/h16_${L}_group_rep$R.err
  python bench.py --depth 24 --only-leg group_fold --steps 20 --chain 10 > $O/h16_${L}_depth24group_rep$R.json 2>> $O/h16_${L}_group_rep$R.err
done; done
unset RTOW_LIB_PATH
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05d/h16_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d.get("value"), d.get("ms_per_step"), d.get("mrays_per_s"))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; tail -c 400 $O/bench_driver_command.json; echo
bash profiles/collect.sh r05 10 > $O/collect_r05.log 2>&1; tail -n 3 $O/collect_r05.log
