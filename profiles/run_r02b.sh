#!/bin/bash
# round-2 second pass: GPU suite with the v5 API (context options, registered host buffers, comm/gather), calibration with both clocks,
# cross-XCD coherence litmus.
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02b
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
timeout 300 python -m pytest tests/test_gpu_comm.py -m gpu -q -rs > $OUT/pytest_comm.log 2>&1
timeout 120 ./profiles/calib/valu_calib 4 > $OUT/calib_w4.json 2> $OUT/calib_w4.err
timeout 120 ./profiles/calib/valu_calib 1 > $OUT/calib_w1.json 2> $OUT/calib_w1.err
timeout 120 ./profiles/calib/coherence_probe > $OUT/coherence.json 2> $OUT/coherence.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/calib_pmc -o calib -- $REPO/profiles/calib/valu_calib 4 > $OUT/calib_pmc.log 2>&1
cd $REPO
tail -5 $OUT/pytest.log; tail -5 $OUT/pytest_comm.log; cat $OUT/coherence.json
