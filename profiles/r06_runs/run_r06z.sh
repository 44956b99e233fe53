# round 6, GPU call z: (1) chains: a wave sets a chunk whose previous batch is not stored aside and takes the next ticket (RTOW_CHAIN_ASIDE); (2) lanes in a hurry + their wave at
# priority 3 (experiment build: RTOW_URGENT_RAISE > 0 = priority on).  Three libraries: without either (nocode), with (1) only (aside), with both (default); same box, two rounds.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
A="24,32,1,32,28,1,3,3,16"; B="16,32,1,24,28,1,4,3,16"
D=$(pwd)/raytracing-in-one-weekend_amd/csrc/build
run() {  # name lib c prio tune
  local NAME=$1 LIB=$2 C=$3 P=$4 T=$5 R=$6
  for L in host_default_adaptive host_default_chain host_default_group; do
    RTOW_LIB_PATH=$LIB RTOW_URGENT_RAISE=$P RTOW_URGENT_RAYS_PER_SAMPLE=$C python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline --tune $T > $O/${L}_${NAME}_$R.json 2> $O/${L}_${NAME}_$R.err
  done
  RTOW_LIB_PATH=$LIB RTOW_URGENT_RAISE=$P RTOW_URGENT_RAYS_PER_SAMPLE=$C python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $T > $O/c2_${NAME}_$R.json 2> $O/c2_${NAME}_$R.err
  RTOW_LIB_PATH=$LIB RTOW_URGENT_RAISE=$P RTOW_URGENT_RAYS_PER_SAMPLE=$C python bench.py --steps 20 --warmup 5 --depth 32 --no-cpu-baseline --no-extras --tune $T > $O/depth32_${NAME}_$R.json 2> $O/depth32_${NAME}_$R.err
  RTOW_LIB_PATH=$LIB RTOW_URGENT_RAISE=$P RTOW_URGENT_RAYS_PER_SAMPLE=$C python bench.py --steps 20 --warmup 5 --depth 32 --spp 50 --chain 1 --no-cpu-baseline --no-extras --tune $T > $O/spp50depth32plain_${NAME}_$R.json 2> $O/spp50depth32plain_${NAME}_$R.err
}
for R in 1 2; do
  run nocode_A $D/librtow_hip_nohurry.so 0 0 $A $R
  run aside_A $D/librtow_hip_aside.so 0 0 $A $R
  run both_off_A "" 0 0 $A $R
  run both_c12_A "" 12 0 $A $R
  run both_c12_prio_A "" 12 1 $A $R
  run both_c8_prio_A "" 8 1 $A $R
  run both_c12_B "" 12 0 $B $R
  run both_c12_prio_B "" 12 1 $B $R
  run both_c8_prio_B "" 8 1 $B $R
  run aside_B $D/librtow_hip_aside.so 0 0 $B $R
done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06z/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); res[os.path.basename(f).rsplit("_", 1)[0]].append((round(d["value"], 1), d["ms_per_step"]))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-44s %s" % (k, res[k]))
PY
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_ties.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tests/soak_chain.py 2>&1 | tail -3
