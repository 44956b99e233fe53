# round 5, GPU call s: PC sampling of the headline launch (rocprofv3 --pc-sampling-beta-enabled), stochastic first, host-trap as the fallback; every step under its own timeout
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05s; mkdir -p $O
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
BENCH="python $REPO/bench.py --steps 10 --warmup 0 --chain 10 --no-cpu-baseline --no-extras"
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --output-format csv -d $O/stoch -o pcs -- $BENCH > $O/stoch.log 2>&1; echo "stochastic rc $?"
ls -la $O/stoch 2>/dev/null | head; find $O/stoch -type f | head
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 --output-format csv -d $O/trap -o pcs -- $BENCH > $O/trap.log 2>&1; echo "host_trap rc $?"
find $O/trap -type f | head
tail -5 $O/stoch.log; tail -5 $O/trap.log
for f in $(find $O -name "*pc_sampling*.csv"); do echo $f; wc -l $f; head -3 $f | cut -c1-400; done
# keep the pulled volume small: compress
for f in $(find $O -name "*.csv" -size +1M); do gzip -9 $f; done
du -sh $O
