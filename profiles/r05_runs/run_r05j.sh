# round 5, GPU call j: what the exact-tie resolver costs the triangle / general kernels (RTOW_CONTEXT_EXACT_TIES_NEVER = the rank-rule kernels, no fix-up): the ceiling of a tie watch for those kinds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
for R in 1 2; do for S in mesh mixed textured; do for F in 0 2; do
  ST=8; [ $S = mesh ] || ST=20
  python bench.py --scene $S --steps $ST --warmup 4 --no-cpu-baseline --no-extras --context-flags $F > $O/${S}_flags${F}_$R.json 2> $O/${S}_flags${F}_$R.err
  python - $O/${S}_flags${F}_$R.json $S $F <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "flags", sys.argv[3], d["value"], d["ms_per_step"], flush=True)
except Exception as e:
    print(sys.argv[2], "flags", sys.argv[3], "FAILED", e, flush=True)
PY
done; done; done
