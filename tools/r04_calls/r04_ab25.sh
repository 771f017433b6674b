#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
DGS_LIB_PATH=$PWD/variants/libdgs_rednat.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "backward or tile_cull or fuzz or variants or huge" > $OUT/c25_parity.log 2>&1; tail -2 $OUT/c25_parity.log
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_rednat.so > $OUT/c25_ab.log 2>&1
cat $OUT/c25_ab.log
