#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "pd_fast_path or view_matrix_gradient_flat" > $OUT/c12_newtests.log 2>&1; echo "rc $?" >> $OUT/c12_newtests.log; grep -v "^$" $OUT/c12_newtests.log | tail -8
