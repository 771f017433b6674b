#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -k "drop_an_overflowed" > $OUT/c23_tests.log 2>&1; tail -30 $OUT/c23_tests.log
