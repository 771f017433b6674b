#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests/test_gpu_train.py -x -q > $OUT/c31_gpu_train.log 2>&1; echo "rc $?" >> $OUT/c31_gpu_train.log; tail -25 $OUT/c31_gpu_train.log
