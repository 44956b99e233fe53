# round 5, GPU call g: is there a better threshold set than the three measured families for the scenes that are not the headline?  (mesh, 10 000 spheres, moving + defocus)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
run() { # name, bench args..., tune
  local name=$1; shift; local tune=$1; shift
  python bench.py "$@" --no-cpu-baseline --no-extras --tune $tune > $O/${name}_$tune.json 2> $O/${name}_$tune.err
  python - $O/${name}_$tune.json $name $tune <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "tune", sys.argv[3], d["value"], d["ms_per_step"], d["config"]["threshold_set"], flush=True)
except Exception as e:
    print(sys.argv[2], "tune", sys.argv[3], "FAILED", e, flush=True)
PY
}
for T in 0,0,0,0,0,0,0,0,0 16,48,1,1,1,1,3,1,24 8,48,1,1,8,1,3,1,24 8,48,1,16,8,1,3,1,24 8,48,1,32,8,1,3,1,24 8,48,8,16,8,1,3,1,24 8,56,1,16,16,1,3,1,24 8,40,1,16,8,1,3,1,24 8,48,1,16,8,1,2,1,24 8,48,1,16,8,1,4,1,24 8,48,1,16,8,1,3,1,32; do
  run mesh $T --scene mesh --steps 8 --warmup 4
done
for T in 0,0,0,0,0,0,0,0,0 24,32,1,32,28,1,3,1,16 24,40,1,32,28,1,3,1,16 24,32,1,40,28,1,3,1,16 16,32,1,32,28,1,3,1,16 32,32,1,32,28,1,3,1,16 24,32,1,24,28,1,3,1,16 24,32,1,32,36,1,3,1,16 24,32,1,32,20,1,3,1,16 24,24,1,32,28,1,3,1,16 24,32,1,32,28,1,3,1,12 24,32,1,32,28,1,3,1,20; do
  run c5 $T --config 5 --steps 10 --warmup 5
done
for T in 0,0,0,0,0,0,0,0,0 24,32,1,32,28,1,3,1,16 24,40,1,32,28,1,3,1,16 24,32,1,40,28,1,3,1,16 16,32,1,32,28,1,3,1,16 24,32,1,24,28,1,3,1,16 24,32,1,32,28,1,4,1,16 24,32,1,32,28,1,3,1,20 24,32,1,32,28,1,3,1,24; do
  run c4 $T --config 4 --steps 10 --warmup 5
done
