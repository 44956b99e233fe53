#!/bin/bash
# Same-box A/B of the pixel regrouping (RtowContextOptions.schedulerTune[7] = super-tile side in tiles; 1 = the tiles as they are): the driver's bench command per side,
# alternating, configs 2 / 4 / 5.   bash profiles/ab_regroup.sh <out dir> [sides] [configs] [repeats]
OUT=${1:?out dir}; SIDES=${2:-"1 2 4 8"}; CONFIGS=${3:-"2 4 5"}; REPS=${4:-2}
mkdir -p "$OUT"
for rep in $(seq 1 $REPS); do
  for cfg in $CONFIGS; do
    for side in $SIDES; do
      extras="--no-extras"; [ "$cfg" = "2" ] && extras=""
      python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline $extras --tune 0,0,0,0,0,0,0,$side,0 > "$OUT/c${cfg}_side${side}_rep${rep}.json" 2> "$OUT/c${cfg}_side${side}_rep${rep}.err"
      python - "$OUT/c${cfg}_side${side}_rep${rep}.json" $cfg $side $rep <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    extra = {k: (d[k]["value"] if isinstance(d.get(k), dict) and "value" in d[k] else None) for k in ("plain_batches", "chain2", "group_fold")}
    print("config", sys.argv[2], "side", sys.argv[3], "rep", sys.argv[4], "value", d["value"], "ms", d["ms_per_step"], extra, flush=True)
except Exception as e:
    print("config", sys.argv[2], "side", sys.argv[3], "rep", sys.argv[4], "FAILED", e, flush=True)
PY
    done
  done
done
