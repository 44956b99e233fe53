# round 6, GPU call aj: counters of the twin with the lanes in a hurry - the reference host's configuration as a chain of 10 (kernel <true,0,32,1,0,false,16>)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aj; mkdir -p $O
POST=0 bash profiles/collect.sh r06_hostdefault_chain 10 --only-leg host_default_chain > $O/collect_hostdefault_chain.log 2>&1
python profiles/summarize.py r06_hostdefault_chain > $O/summarize.log 2>&1; tail -3 $O/summarize.log
cp profiles/r06_hostdefault_chain_* $O/ 2>/dev/null; ls $O
