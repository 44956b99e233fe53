# round 5, GPU call af: larger soak of the round's final build (slot tickets included): whole frames at twice the sample counts, 6 000 + 1 600 heavy fuzz seeds, chain soak, group / chain / comm / variant tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05af; mkdir -p $O
timeout 2400 python tests/soak_frames.py 2.0 > $O/soak_frames.log 2>&1; tail -3 $O/soak_frames.log
RTOW_FUZZ_SEEDS=6000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 6 2>&1 | tail -2 > $O/fuzz_6000.log; cat $O/fuzz_6000.log
RTOW_FUZZ_SEEDS=1600 RTOW_FUZZ_HEAVY=1 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 6 2>&1 | tail -2 > $O/fuzz_1600_heavy.log; cat $O/fuzz_1600_heavy.log
timeout 900 python tests/soak_chain.py > $O/soak_chain.log 2>&1; tail -2 $O/soak_chain.log
python -m pytest tests/test_gpu_group.py tests/test_gpu_chain.py tests/test_gpu_comm.py tests/test_gpu_variants.py -q -n 4 2>&1 | tail -2
