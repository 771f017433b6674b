#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1
{
for mode in views subframes; do
  timeout 900 python tools/dist_training_check.py --ranks 2 --mode $mode --iters 300 --graph always --densify-interval 25 --ar-chunks 4 --random-sample 2>&1 | grep -v Gloo
  DGS_DIST_ALLREDUCE=p2p DGS_DIST_P2P_MIN_NUMEL=0 timeout 900 python tools/dist_training_check.py --ranks 2 --mode $mode --iters 150 --graph always --densify-interval 25 --ar-chunks 4 2>&1 | grep -v Gloo
done
} > $OUT/c27_long_dist.log 2>&1
cat $OUT/c27_long_dist.log
