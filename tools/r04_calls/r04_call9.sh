#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1
for rep in 1 2 3 4; do
for g in off auto; do for c in 1 4; do
  echo -n "rep $rep graph $g chunks $c: "; timeout 300 python tools/dist_training_check.py --ranks 2 --mode views --iters 14 --ar-chunks $c --graph $g 2>&1 | grep "params sha1"
done; done; done > $OUT/c9_flaky.log 2>&1
cat $OUT/c9_flaky.log
