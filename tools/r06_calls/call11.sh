#!/bin/bash
# round 6 call 11: T / (1 - alpha) with one residual correction (two more FMAs per backward pass) against the plain
# reciprocal product: parity at the metric size (exact exempt set) and step time, interleaved
mkdir -p gpurun_out/r06
L=gpurun_out/r06/div_refine.log
: > $L
for rep in 1 2 3; do
  echo -n "refine   " >> $L; timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/brief.py >> $L
  echo -n "norefine " >> $L; DGS_LIB_PATH=variants/libdgs_norefine.so timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/brief.py >> $L
done
cat $L
DGS_PARITY_REPORT=1 timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s -k "benchmarked and metric" 2>&1 | grep -E "dL_dconic|FAIL|passed|failed|exempt" | cut -c1-500 | tee gpurun_out/r06/div_refine_parity.log
