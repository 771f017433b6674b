#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 3000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -s > $OUT/c18_margin.log 2>&1; echo "rc $?" >> $OUT/c18_margin.log; grep -i "dL_dconic\|named exc\|passed\|failed\|rc \|unstable" $OUT/c18_margin.log | tail -40
