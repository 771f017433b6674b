#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
tools/addtid_probe > $OUT/c4_probe.log 2>&1; cat $OUT/c4_probe.log
for lib in addtid addtid_occ6; do
DGS_LIB_PATH=$PWD/variants/libdgs_$lib.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "backward_vs_oracle or deterministic or fused_equals or fuzz" > $OUT/c4_${lib}_parity.log 2>&1; tail -2 $OUT/c4_${lib}_parity.log
done
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_addtid.so variants/libdgs_addtid_occ6.so > $OUT/c4_ab.log 2>&1
cat $OUT/c4_ab.log
