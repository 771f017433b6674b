#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/c7_parity.log 2>&1; echo "rc $?" >> $OUT/c7_parity.log; tail -3 $OUT/c7_parity.log
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_al64.so variants/libdgs_al32.so variants/libdgs_alall.so > $OUT/c7_ab.log 2>&1
cat $OUT/c7_ab.log
