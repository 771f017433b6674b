#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "backward or fuzz or tile_cull or variants or deterministic or fused_equals or huge or ragged or all_culled" > $OUT/c15_parity.log 2>&1; echo "rc $?" >> $OUT/c15_parity.log; tail -3 $OUT/c15_parity.log
timeout 1500 python tools/ab_bench.py --steps 30 variants/libdgs_base.so default > $OUT/c15_ab.log 2>&1
cat $OUT/c15_ab.log
