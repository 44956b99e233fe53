#!/bin/bash
# round-3 pass ae: wide-code kernels of the volume kinds (variants + the 81 925-entity fog mesh), finalize with nine conversions in flight (parity over
# 2^32 operands, post-pass tests, achieved GB/s at 1080p and 4K)
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03ae
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_variants.py -q -x -k "wide_code" > $OUT/variants.log 2>&1; tail -3 $OUT/variants.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "fog or mesh_grid" > $OUT/fog.log 2>&1; tail -3 $OUT/fog.log
timeout 600 python -m pytest tests/test_gpu_detmath.py tests/test_gpu_api.py -q -x -k "finalize or combine or post or metrics" > $OUT/post.log 2>&1; tail -3 $OUT/post.log
for rep in 1 2 3; do
  for size in 1920x1080 3840x2160; do timeout 300 python bench.py --post-only $size > $OUT/post_${size}_$rep.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/post_${size}_$rep.json'))['post_passes']['$size']; print('$size', {k: v['GBps'] for k, v in d.items()})"; done
done
