#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
for lib in default rp4; do
  if [ "$lib" != "default" ]; then export DGS_LIB_PATH=$PWD/variants/libdgs_$lib.so; else unset DGS_LIB_PATH; fi
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "backward or tile_cull_gradients or fuzz or huge" > $OUT/c34_parity_$lib.log 2>&1; echo "$lib: $(tail -1 $OUT/c34_parity_$lib.log)"
done
unset DGS_LIB_PATH
timeout 1500 python tools/ab_bench.py --steps 30 variants/libdgs_base.so default variants/libdgs_rp2.so variants/libdgs_rp4.so > $OUT/c34_ab.log 2>&1
cat $OUT/c34_ab.log
