# round 5, GPU call c: comm tests incl. the real RCCL loop-back, stage statistics for the regrouping, rocprofv3 evidence for C4 / C5 / mesh / host default / group launches,
# the adaptive host schedule with and without the chunk order
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
python -m pytest tests/test_gpu_comm.py -x -q 2>&1 | tail -15 > $O/tests_comm.log; cat $O/tests_comm.log
STATS=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so
for T in 1 8 40; do
  RTOW_LIB_PATH=$STATS python bench.py --steps 1 --warmup 0 --chain 1 --no-cpu-baseline --no-extras --tune 0,0,0,0,0,0,0,$T,0 > $O/stats_tune$T.json 2> $O/stats_tune$T.log
  RTOW_LIB_PATH=$STATS python bench.py --steps 2 --warmup 0 --chain 1 --no-cpu-baseline --no-extras --tune 0,0,0,0,0,0,0,$T,0 > $O/stats2_tune$T.json 2> $O/stats2_tune$T.log
done
for F in 0 16; do for R in 1 2; do
  python bench.py --only-leg host_default_adaptive --steps 20 --context-flags $F > $O/adaptive_flags${F}_rep$R.json 2> $O/adaptive_flags${F}_rep$R.err; tail -c 900 $O/adaptive_flags${F}_rep$R.json; echo
done; done
python bench.py --only-leg host_default_group --steps 20 --chain 10 > $O/hostdefault_group.json 2> $O/hostdefault_group.err; tail -c 600 $O/hostdefault_group.json; echo
POST=0 L2=1 bash profiles/collect.sh r05_c4 10 --config 4 > $O/collect_c4.log 2>&1
POST=0 bash profiles/collect.sh r05_c5 10 --config 5 > $O/collect_c5.log 2>&1
POST=0 L2=1 bash profiles/collect.sh r05_mesh 4 --scene mesh > $O/collect_mesh.log 2>&1
POST=0 bash profiles/collect.sh r05_hostdefault 10 --only-leg host_default_group > $O/collect_hostdefault.log 2>&1
POST=0 bash profiles/collect.sh r05_group 10 --only-leg group_fold > $O/collect_group.log 2>&1
tail -n 2 $O/collect_*.log
