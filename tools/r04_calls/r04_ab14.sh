#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
DGS_LIB_PATH=$PWD/variants/libdgs_st512.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "radix_sort or binning" > $OUT/c14_st512.log 2>&1; tail -2 $OUT/c14_st512.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "radix_sort or binning or tile_cull" > $OUT/c14_default.log 2>&1; tail -2 $OUT/c14_default.log
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_st512.so variants/libdgs_cw1.so variants/libdgs_cw2.so > $OUT/c14_ab.log 2>&1
cat $OUT/c14_ab.log
