# round 5, GPU call aa: soak of the session's final build (node loads without flat_load, launch constants on use / from LDS) - whole frames at higher sample counts against the oracle (every pixel), fuzz with 3 000 seeds (+ 800 heavy), chain soak
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05aa; mkdir -p $O
timeout 1500 python tests/soak_frames.py 1.0 > $O/soak_frames.log 2>&1; tail -25 $O/soak_frames.log
RTOW_FUZZ_SEEDS=3000 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 6 2>&1 | tail -4 > $O/fuzz_3000.log; cat $O/fuzz_3000.log
RTOW_FUZZ_SEEDS=800 RTOW_FUZZ_HEAVY=1 timeout 1800 python -m pytest tests/test_gpu_fuzz.py -q -n 6 2>&1 | tail -4 > $O/fuzz_800_heavy.log; cat $O/fuzz_800_heavy.log
timeout 900 python tests/soak_chain.py > $O/soak_chain.log 2>&1; tail -8 $O/soak_chain.log
