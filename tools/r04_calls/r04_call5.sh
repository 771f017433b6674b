#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
bash tools/r04_calls/r04_valu.sh > $OUT/c5_valu.log 2>&1; tail -40 $OUT/c5_valu.log
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_addtid.so variants/libdgs_addtid_nw.so > $OUT/c5_ab.log 2>&1
cat $OUT/c5_ab.log
