#!/bin/bash
# round-3 pass ag: finalize with integer zone check / clamp modifier (165 VALU instructions per pixel, was 240): parity over 2^32 operands x 9 positions, post-pass tests, GB/s
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ag
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_detmath.py tests/test_gpu_api.py -q -x -k "finalize or combine or post or metrics" > $OUT/post.log 2>&1; tail -3 $OUT/post.log
tests/build/finalize_parity > $OUT/finalize_parity.txt 2>&1; cat $OUT/finalize_parity.txt
for rep in 1 2 3; do
  for size in 1920x1080 3840x2160; do timeout 300 python bench.py --post-only $size > $OUT/post_${size}_$rep.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/post_${size}_$rep.json'))['post_passes']['$size']; print('$size', {k: v['GBps'] for k, v in d.items()})"; done
done
