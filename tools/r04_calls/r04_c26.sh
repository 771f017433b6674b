#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests/test_gpu_train.py -x -q -k "two_ranks or rccl" > $OUT/c26_tests.log 2>&1; tail -3 $OUT/c26_tests.log
