#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_nopost.so variants/libdgs_nomisched.so > $OUT/c30_ab.log 2>&1
cat $OUT/c30_ab.log
