# round 5, GPU call o: the final build once more through the whole -m gpu suite, and the counter summaries of every kernel family re-collected on it (a tile's tickets most expensive first)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
python -m pytest tests -q -m gpu -n 4 2>&1 | tail -4 > $O/tests_gpu.log; cat $O/tests_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
POST=0 L2=1 bash profiles/collect.sh r05_c4 10 --config 4 > $O/collect_c4.log 2>&1
POST=0 bash profiles/collect.sh r05_c5 10 --config 5 > $O/collect_c5.log 2>&1
POST=0 L2=1 bash profiles/collect.sh r05_mesh 4 --scene mesh > $O/collect_mesh.log 2>&1
POST=0 bash profiles/collect.sh r05_hostdefault 10 --only-leg host_default_group > $O/collect_hostdefault.log 2>&1
POST=0 bash profiles/collect.sh r05_group 10 --only-leg group_fold > $O/collect_group.log 2>&1
for f in $O/collect_*.log; do echo $f; tail -n 2 $f; done
