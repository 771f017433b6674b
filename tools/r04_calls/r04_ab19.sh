#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_ntld.so variants/libdgs_ntst.so variants/libdgs_ntboth.so variants/libdgs_ntrows.so > $OUT/c19_ab.log 2>&1
cat $OUT/c19_ab.log
