#!/bin/bash
# Flakiness hunt on one more box: the N-rank tests five times over, then the whole suite again.
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
for i in 1 2 3 4 5; do timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -k "two_ranks or rccl or captured" > $OUT/c29_dist_$i.log 2>&1; echo "dist round $i: $(tail -1 $OUT/c29_dist_$i.log)"; done
timeout 3000 python -m pytest tests/ -x -q -m gpu > $OUT/c29_suite.log 2>&1; echo "suite: $(tail -1 $OUT/c29_suite.log)"
