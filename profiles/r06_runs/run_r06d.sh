# round 6, GPU call d: (1) pixel boundaries in company - K lanes of a wave must want a boundary before the boundary block runs (schedulerTune[7] bits 12..15), one library, same box,
# two alternating rounds: headline chains, host-default groups / adaptive, group_fold, per-sample plain launches;  (2) the mesh kernel's vector-memory path: TA / TCP counters.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1      # the box's first run
T="24,32,1,32,28,1,3"
for R in 1 2; do for K in 1 2 3 4 6; do
  P=$((3 + 4096 * K)); TUNE="$T,$P,16"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $TUNE > $O/c2_k${K}_$R.json 2> $O/c2_k${K}_$R.err
  for L in host_default_group host_default_adaptive group_fold; do
    python bench.py --only-leg $L --chain 10 --steps 20 --no-cpu-baseline --tune $TUNE > $O/${L}_k${K}_$R.json 2> $O/${L}_k${K}_$R.err
  done
  python bench.py --rng per-sample --chain 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --tune $TUNE > $O/persample_k${K}_$R.json 2> $O/persample_k${K}_$R.err
  python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune $TUNE > $O/c5_k${K}_$R.json 2> $O/c5_k${K}_$R.err
done; done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r06d/*_k*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(round(d["value"], 1))
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-36s %s" % (k, res[k]))
PY
# (2) counters of the mesh launch: one chain of 4 batches, the counter passes read the last dispatch
REPO=$(pwd); P=$REPO/gpurun_out/r06d/prof_mesh; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
ONE="python $REPO/bench.py --scene mesh --steps 4 --warmup 0 --chain 4 --no-cpu-baseline --no-extras --prewarm 0"
pmc() { local name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $P/pmc_$name -o bench -- $ONE > $P/pmc_$name.log 2>&1 || echo "pass $name failed" >> $P/failed_passes.txt; }
pmc ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pmc ta2 TA_BUSY_avr TA_BUSY_max TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pmc tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pmc tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
pmc tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
cd $REPO
python - <<'PY'
import csv, glob, os, json
out = {}
for d in sorted(glob.glob("gpurun_out/r06d/prof_mesh/pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "sample_batch_kernel" in r.get("Kernel_Name", "")]
        if not rows: continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last: out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
json.dump(out, open("gpurun_out/r06d/mesh_ta_tcp_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat $P/failed_passes.txt 2>/dev/null
