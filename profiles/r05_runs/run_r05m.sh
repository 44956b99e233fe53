# round 5, GPU call m: the order of a tile's tickets by cost LEVELS (pixels of one level keep their row order): speed and HBM traffic against the exact sort and the row order
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05m; mkdir -p $O
python -m pytest tests/test_gpu_regroup.py -q -x -n 4 2>&1 | tail -3
for R in 1 2 3; do for T in 1 3 19 35 51; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune 0,0,0,0,0,0,0,$T,0 > $O/c2_tune${T}_$R.json 2> $O/c2_tune${T}_$R.err
  python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --tune 0,0,0,0,0,0,0,$T,0 > $O/c5_tune${T}_$R.json 2> $O/c5_tune${T}_$R.err
done; done
python - <<'PY'
import json, glob, os, collections
res = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r05m/c*_tune*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); name = os.path.basename(f).rsplit("_", 1)[0]
        res[name].append(d["value"])
        for k in ("plain_batches", "chain2", "group_fold"):
            if k in d: res[name + " " + k].append(d[k]["value"])
    except Exception as e:
        print(f, "FAILED", e)
for k in sorted(res): print("%-40s %s  mean %.0f" % (k, res[k], sum(res[k]) / len(res[k])))
PY
cd /tmp && export TMPDIR=/tmp
for T in 1 3 19 35 51; do
  ONE="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 0 --chain 10 --no-cpu-baseline --no-extras --tune 0,0,0,0,0,0,0,$T,0"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -o bench -- $ONE > $O/pmc_fetch_$T.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -o bench -- $ONE > $O/pmc_write_$T.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for T in (1, 3, 19, 35, 51):
    tot = {}
    for what in ("fetch", "write"):
        for f in glob.glob("gpurun_out/r05m/pmc_%s_%d/**/*counter_collection.csv" % (what, T), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "sample_batch_kernel" in r["Kernel_Name"]]
            full = max(int(r["Grid_Size"]) for r in rows); rows = [r for r in rows if int(r["Grid_Size"]) == full]
            last = max(int(r["Dispatch_Id"]) for r in rows)
            tot[what] = sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last)
    print("tune", T, "read GB (x2)", tot.get("fetch", 0) * 1024 * 2 / 1e9, "written GB", tot.get("write", 0) * 1024 / 1e9, "total", (tot.get("fetch", 0) * 2 + tot.get("write", 0)) * 1024 / 1e9)
PY
