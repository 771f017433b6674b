#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -s -k "beyond_27 or pd_fast_path" > $OUT/c24_tests.log 2>&1; grep -v "^$" $OUT/c24_tests.log | tail -12
