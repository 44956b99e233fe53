# round 5, GPU call u: stage statistics build (profiles/experiments/build.sh stats -DRTOW_STATS) on the headline as chains of 10 and groups of 10 (the SECOND launch of each run: the first carries the context's probes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05u; mkdir -p $O
export RTOW_LIB_PATH=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so
python bench.py --steps 20 --warmup 0 --chain 10 --no-cpu-baseline --no-extras > $O/chain10.json 2> $O/chain10.err
python bench.py --steps 20 --warmup 0 --chain 10 --no-cpu-baseline --no-extras --only-leg group_fold > $O/group10.json 2> $O/group10.err
for f in chain10 group10; do echo "== $f"; grep -c "regen lanes" $O/$f.err; grep "^\[stats\]" $O/$f.err | grep -v "wave #" | tail -26; done
