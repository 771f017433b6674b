#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
for lib in urow_spec2; do
DGS_LIB_PATH=$PWD/variants/libdgs_$lib.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "backward_vs_oracle or deterministic or fused_equals or fuzz" > $OUT/c6_${lib}_parity.log 2>&1; tail -2 $OUT/c6_${lib}_parity.log
done
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_addtid.so variants/libdgs_urow.so variants/libdgs_spec2.so variants/libdgs_urow_spec2.so > $OUT/c6_ab.log 2>&1
cat $OUT/c6_ab.log
