cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ad; mkdir -p $O
export RTOW_LIB_PATH=raytracing-in-one-weekend_amd/csrc/build/librtow_hip_stats.so
python bench.py --config 5 --steps 20 --warmup 0 --chain 10 --no-cpu-baseline --no-extras > $O/c5.json 2> $O/c5.err
python bench.py --steps 20 --warmup 0 --chain 10 --no-cpu-baseline --no-extras > $O/c2.json 2> $O/c2.err
python bench.py --config 4 --steps 20 --warmup 0 --chain 10 --no-cpu-baseline --no-extras > $O/c4.json 2> $O/c4.err
for f in c2 c5 c4; do echo "== $f"; grep "^\[stats\]" $O/$f.err | grep -v "wave #" | tail -26 | head -24; done
