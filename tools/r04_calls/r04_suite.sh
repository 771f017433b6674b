#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 3300 python -m pytest tests/ -x -q -m gpu --durations=15 > $OUT/gpu_suite_${1:-a}.log 2>&1; echo "rc $?" >> $OUT/gpu_suite_${1:-a}.log; tail -25 $OUT/gpu_suite_${1:-a}.log
