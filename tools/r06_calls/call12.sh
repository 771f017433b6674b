#!/bin/bash
# round 6 call 12: the conic with exact scalings + one multiply by log2(e) per pass (new default) against the log2-domain
# conic of rounds 2-5: step time (interleaved) and parity at the metric size with the exact exempt set
mkdir -p gpurun_out/r06
L=gpurun_out/r06/log2conic.log
: > $L
for rep in 1 2 3; do
  echo -n "exact-coefficients " >> $L; timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/brief.py >> $L
  echo -n "log2-conic (r2-r5) " >> $L; DGS_LIB_PATH=variants/libdgs_log2conic.so timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/brief.py >> $L
done
cat $L
DGS_PARITY_REPORT=1 timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "benchmarked and metric" 2>&1 | grep -E "'dL_dconic'|'dL_dmeans2D'|'xyz'|'scaling'|FAIL|passed|failed|exempt" | cut -c1-600 | tee gpurun_out/r06/log2conic_parity.log
