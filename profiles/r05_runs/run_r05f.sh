# round 5, GPU call f: how far a "most populated stage only" schedule is from the shipped thresholds (thresholds of 64/64 are never met, so every second trip forces the fullest stage),
# and the stage statistics of the mesh kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
for T in 0,0,0,0,0,0,0,0,0 64,64,64,64,64,64,3,1,16 48,48,48,48,48,48,3,1,16 32,48,1,48,48,1,3,1,16 24,32,1,32,28,1,3,1,16 24,40,8,32,28,1,3,1,16; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $T > $O/tune_$T.json 2> $O/tune_$T.err
  python - $O/tune_$T.json $T <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("tune", sys.argv[2], d["value"], d["ms_per_step"], d["config"]["scheduler_tune"], d["config"]["threshold_set"], flush=True)
except Exception as e:
    print("tune", sys.argv[2], "FAILED", e, flush=True)
PY
done
STATS=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so
RTOW_LIB_PATH=$STATS python bench.py --scene mesh --steps 1 --warmup 0 --chain 1 --no-cpu-baseline --no-extras > $O/stats_mesh.json 2> $O/stats_mesh.log; grep "\[stats\]" $O/stats_mesh.log | head -24
