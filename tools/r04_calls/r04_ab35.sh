#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
DGS_LIB_PATH=$PWD/variants/libdgs_asmread.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "backward or tile_cull_gradients or fuzz or huge or deterministic" > $OUT/c35_parity.log 2>&1; echo "asmread: $(tail -1 $OUT/c35_parity.log)"
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_asmread.so > $OUT/c35_ab.log 2>&1
cat $OUT/c35_ab.log
