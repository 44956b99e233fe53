# round 6, GPU call ae: the bound of the lanes in a hurry is set only for launches a twin serves (r06ad's driver line: the per-sample policies read it as a pixel gate of millions and
# ran at 8 311 instead of 9 170).  The driver's command with its secondary legs, under a kernel trace too (which variant served which leg), smoke, the GPU suite.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ae; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
for R in 1 2; do
  python bench.py --rng per-sample --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/persample_$R.json 2> $O/persample_$R.err
  for L in host_default_adaptive host_default_chain host_default_group plain_two_in_flight group_fold; do
    python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline > $O/${L}_$R.json 2> $O/${L}_$R.err
  done
done
REPO=$(pwd); (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace -o bench -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $REPO/$O/trace.log 2>&1)
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/driver_command_kernel_stats.csv \;
rm -rf $O/trace
python - <<'PY'
import json, glob, os
d = json.loads(open("gpurun_out/r06ae/bench_driver_command.json").read().strip().splitlines()[-1])
print("driver", d["value"], "per_sample", d.get("per_sample", {}).get("value"), "xoro", d.get("per_sample_xoroshiro", {}).get("value"), "host_default", {k: v for k, v in d.get("host_default", {}).items() if k in ("chain", "group_fold", "adaptive")})
for f in sorted(glob.glob("gpurun_out/r06ae/*_[12].json")):
    print(os.path.basename(f), json.loads(open(f).read().strip().splitlines()[-1])["value"])
PY
grep sample_batch_kernel $O/driver_command_kernel_stats.csv | cut -d, -f1-4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
