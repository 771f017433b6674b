#!/bin/bash
# Round 4 A/B of the compositing backward: occupancy 6 (<= 80 VGPRs), 15-way hit-mask specialisation, ds_write_addtid_b32.
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
for lib in addtid_occ6 addtid_spec; do
  DGS_LIB_PATH=$PWD/variants/libdgs_$lib.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "backward_vs_oracle or deterministic or fused_equals" > $OUT/ab_parity_$lib.log 2>&1
  echo "$lib parity rc $?" >> $OUT/ab_parity_$lib.log
  tail -3 $OUT/ab_parity_$lib.log
done
timeout 1500 python tools/ab_bench.py --steps 30 default variants/libdgs_occ6.so variants/libdgs_spec.so variants/libdgs_addtid.so variants/libdgs_addtid_occ6.so variants/libdgs_addtid_spec.so > $OUT/ab_composite.log 2>&1
cat $OUT/ab_composite.log
