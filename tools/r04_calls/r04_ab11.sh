#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "pd_fast_path or view_matrix_gradient_flat" > $OUT/c11_newtests.log 2>&1; echo "rc $?" >> $OUT/c11_newtests.log; grep -v "^$" $OUT/c11_newtests.log | tail -12
DGS_LIB_PATH=$PWD/variants/libdgs_firstq.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "backward_vs_oracle or deterministic or fused_equals or fuzz or tile_cull_gradients" > $OUT/c11_firstq_parity.log 2>&1; tail -2 $OUT/c11_firstq_parity.log
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_firstq.so variants/libdgs_firstq_w6.so > $OUT/c11_ab.log 2>&1
cat $OUT/c11_ab.log
