# round 5, GPU call ai: twelve chain-of-10 launches in one process under one counter pass (GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_INSTS_VALU): do the slow launches take more cycles or run at a lower clock?
REPO=$GRAFT_REPO_ROOT; O=$REPO/gpurun_out/r05ai; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc -o run -- python $REPO/bench.py --steps 160 --warmup 0 --chain 10 --prewarm 0 --no-cpu-baseline --no-extras > $O/run.log 2>&1
python - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "sample_batch_kernel" not in r["Kernel_Name"]: continue
    k = int(r["Dispatch_Id"]); rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
    for c in ("Start_Timestamp", "End_Timestamp"):
        if c in r: rows[k][c] = int(r[c])
for k in sorted(rows):
    d = rows[k]
    dur = (d.get("End_Timestamp", 0) - d.get("Start_Timestamp", 0)) / 1e6
    if dur < 100: continue
    print("dispatch %4d  %8.2f ms  GUI_ACTIVE %.4g  -> %.0f MHz   VALU %.5g  BUSY %.4g" % (k, dur, d.get("GRBM_GUI_ACTIVE", 0), d.get("GRBM_GUI_ACTIVE", 0) / dur / 1e3, d.get("SQ_INSTS_VALU", 0), d.get("SQ_BUSY_CYCLES", 0)))
PY
