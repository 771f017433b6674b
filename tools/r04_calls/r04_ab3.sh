#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "depth_order or binning or tile_cull or fused_equals or forward" > $OUT/c3_depth_order_tests.log 2>&1; echo "rc $?" >> $OUT/c3_depth_order_tests.log; tail -4 $OUT/c3_depth_order_tests.log
DGS_LIB_PATH=$PWD/variants/libdgs_ds256.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "depth_order" > $OUT/c3_ds256_tests.log 2>&1; tail -2 $OUT/c3_ds256_tests.log
DGS_LIB_PATH=$PWD/variants/libdgs_addtid.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "backward_vs_oracle or deterministic or fused_equals" > $OUT/c3_addtid_parity.log 2>&1; tail -2 $OUT/c3_addtid_parity.log
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_addtid.so variants/libdgs_ds256.so > $OUT/c3_ab.log 2>&1
cat $OUT/c3_ab.log
