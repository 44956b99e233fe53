# round 6, GPU call c: the timing-sensitive cancellation tests alone, the new tie / comm tests, then the whole suite again (4 workers)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_api.py::test_cancellation_token tests/test_gpu_chain.py::test_cancelled_chain_returns_promptly_and_leaves_the_context_usable -q > $O/cancel.log 2>&1; tail -3 $O/cancel.log
timeout 900 python -m pytest tests/test_gpu_ties.py tests/test_gpu_comm.py -q -x > $O/ties_comm.log 2>&1; tail -5 $O/ties_comm.log
timeout 1500 python -m pytest tests -m gpu -q -n 4 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
