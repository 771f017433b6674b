#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1
for rep in 1 2 3; do
for g in off always; do for c in 1 4; do
  echo -n "subframes rep $rep graph $g chunks $c: "; timeout 300 python tools/dist_training_check.py --ranks 2 --mode subframes --iters 20 --ar-chunks $c --graph $g --random-sample 2>&1 | grep "params sha1"
done; done; done > $OUT/c32_matrix_subframes.log 2>&1
cat $OUT/c32_matrix_subframes.log
unset DGS_DIST_BACKEND DGS_DIST_ONE_DEVICE
timeout 600 python tools/rccl_smoke.py 2>&1 | grep "rccl smoke"
