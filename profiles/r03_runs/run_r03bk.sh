#!/bin/bash
# round-3 pass bk: bench.py --gpus 8 end to end in the debug mode (8 ranks sharing the one GPU, test transport): the N = 8 code path of the driver's scaling run
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03bk
rm -rf $OUT; mkdir -p $OUT
export RTOW_BENCH_DEBUG_SHARED_GPU=1
for n in 8 4; do
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 4 --warmup 1 ) > $OUT/bench_$n.json 2> $OUT/bench_$n.err
python -c "
import json; d=json.loads([l for l in open('$OUT/bench_$n.json') if l.startswith('{')][-1]); print('n=$n', d['value'], d['n_gpus'], d['scaling'], d['ms_per_step'], d['config']['partition'][:80], d['config']['gather'][:60], list(d.get('partitions', {}).keys()))" || tail -5 $OUT/bench_$n.err
grep real $OUT/bench_$n.err
done
