# round 6, GPU call j: rocprofv3 evidence of the final build (kernel trace + PMC passes of every kernel family, post passes), and the partition emulation on the r06 kernel.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
bash profiles/collect.sh r06 10 > $O/collect_c2.log 2>&1; tail -n 2 $O/collect_c2.log
POST=0 bash profiles/collect.sh r06_hostdefault 10 --only-leg host_default_group > $O/collect_hostdefault.log 2>&1
POST=0 bash profiles/collect.sh r06_group 10 --only-leg group_fold > $O/collect_group.log 2>&1
POST=0 L2=1 bash profiles/collect.sh r06_c4 10 --config 4 > $O/collect_c4.log 2>&1
POST=0 bash profiles/collect.sh r06_c5 10 --config 5 > $O/collect_c5.log 2>&1
POST=0 L2=1 bash profiles/collect.sh r06_mesh 4 --scene mesh > $O/collect_mesh.log 2>&1
for f in $O/collect_*.log; do echo $f; tail -n 2 $f; done
timeout 900 python profiles/emulate_partitions.py --config 2 > $O/partitions_c2.json 2> $O/partitions_c2.err; tail -c 600 $O/partitions_c2.json
timeout 1200 python profiles/emulate_partitions.py --config 3 > $O/partitions_c3.json 2> $O/partitions_c3.err; tail -c 600 $O/partitions_c3.json
