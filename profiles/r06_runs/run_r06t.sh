# round 6, GPU call t: the stage thresholds were measured on the headline kernels (depth 8, 256 spp).  Do the reference host's committed configuration (depth 32, 16-byte records,
# 50 spp: kernel <true,0,32,1,...>) and the per-sample policy want others?  One library, --tune, same box, one value moved at a time, baseline interleaved.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
I=0
for T in "24,32,1,32,28,1,3,3,16" "16,32,1,32,28,1,3,3,16" "32,32,1,32,28,1,3,3,16" "40,32,1,32,28,1,3,3,16" "24,24,1,32,28,1,3,3,16" "24,40,1,32,28,1,3,3,16" "24,48,1,32,28,1,3,3,16" \
         "24,32,1,32,28,1,3,3,16" "24,32,16,32,28,1,3,3,16" "24,32,1,16,28,1,3,3,16" "24,32,1,24,28,1,3,3,16" "24,32,1,40,28,1,3,3,16" "24,32,1,32,16,1,3,3,16" "24,32,1,32,20,1,3,3,16" \
         "24,32,1,32,36,1,3,3,16" "24,32,1,32,28,1,3,3,16" "24,32,1,32,28,1,2,3,16" "24,32,1,32,28,1,4,3,16" "24,32,1,32,28,1,3,3,12" "24,32,1,32,28,1,3,3,24" "32,40,1,40,36,1,3,3,16" \
         "24,32,1,32,28,1,3,3,16"; do
  I=$((I + 1)); N=$(printf "%02d" $I)_$(echo $T | tr ',' '_')
  for L in host_default_group host_default_adaptive; do
    python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline --tune $T > $O/${L}_$N.json 2> $O/${L}_$N.err
  done
  python bench.py --rng per-sample --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --tune $T > $O/persample_$N.json 2> $O/persample_$N.err
done
python - <<'PY'
import json, glob, os, collections
for leg in ("host_default_group", "host_default_adaptive", "persample"):
    print(leg)
    for f in sorted(glob.glob("gpurun_out/r06t/%s_*.json" % leg)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1]); print("   %-40s %.1f" % (os.path.basename(f)[len(leg) + 1:-5], d["value"]))
        except Exception as e:
            print(f, "FAILED", e)
PY
