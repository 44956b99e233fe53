#!/bin/bash
# round-2 first measurement pass (run through gpurun from the repo root): GPU suite, the four single-GPU configs through bench.py,
# VALU calibration (plain + under the PMC counters it pins), tile-split emulation of configs 2 and 3 under both RNG policies.
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02a
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
for c in 2 4 5 3; do
  timeout 300 python bench.py --config $c --steps 5 --warmup 1 > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err
done
timeout 120 ./profiles/calib/valu_calib 4 > $OUT/calib_w4.json 2> $OUT/calib_w4.err
timeout 120 ./profiles/calib/valu_calib 1 > $OUT/calib_w1.json 2> $OUT/calib_w1.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/calib_pmc -o calib -- $REPO/profiles/calib/valu_calib 4 > $OUT/calib_pmc.log 2>&1
cd $REPO
for c in 2 3; do for r in reference per-sample; do
  timeout 600 python profiles/emulate_tile_split.py --config $c --rng $r > $OUT/tiles_c${c}_$r.json 2> $OUT/tiles_c${c}_$r.err
done; done
tail -3 $OUT/pytest.log; cat $OUT/bench_c*.json | cut -c1-400; ls -la $OUT
