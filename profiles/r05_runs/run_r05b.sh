cd /tmp && export TMPDIR=/tmp; rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r05b_counters.txt 2>&1; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_probe.py tests/test_gpu_regroup.py -x -q 2>&1 | tail -15 > gpurun_out/r05b_tests.log; cat gpurun_out/r05b_tests.log
bash profiles/ab_regroup.sh gpurun_out/r05b "1 20 24 36 40" "2 4 5" 1 2>&1 | tee gpurun_out/r05b_ab.log
POST=0 bash profiles/collect.sh r05b_side1 10 --tune 0,0,0,0,0,0,0,1,0 > gpurun_out/r05b_collect1.log 2>&1
POST=0 bash profiles/collect.sh r05b_side8 10 --tune 0,0,0,0,0,0,0,8,0 > gpurun_out/r05b_collect8.log 2>&1
tail -3 gpurun_out/r05b_collect1.log gpurun_out/r05b_collect8.log
