#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 3000 python -m pytest tests/test_gpu_configs.py -x -q -s > $OUT/r04_gpu_configs.log 2>&1; echo "rc $?" >> $OUT/r04_gpu_configs.log; grep -i "conic\|passed\|failed\|rc " $OUT/r04_gpu_configs.log | tail -30
