# round 5, GPU call t: the per-sample policies (north_star's lane-per-sample shape, not the reference stream) against the reference stream, plain launches and chains, with a kernel trace of the per-sample run
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05t; mkdir -p $O
for V in reference per-sample per-sample-xoroshiro; do for CH in 1 10; do
  python bench.py --rng $V --steps 20 --warmup 5 --chain $CH --no-cpu-baseline --no-extras > $O/${V}_chain$CH.json 2> $O/${V}_chain$CH.err
  python - $O/${V}_chain$CH.json $V $CH <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "chain", sys.argv[3], d["value"], d["ms_per_step"], d.get("rays_per_sample"), d.get("kernel_ms_per_step"))
except Exception as e: print(sys.argv[2], sys.argv[3], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
done; done
REPO=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o ps -- python $REPO/bench.py --rng per-sample --steps 10 --warmup 0 --chain 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
cat $(find $O/trace -name "*kernel_stats.csv") | cut -c1-200 | head -12
