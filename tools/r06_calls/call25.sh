#!/bin/bash
# round 6 call 25: the bit-exact stage statements (preprocess bits, duplicate offsets, keys, point lists, ranges; tile-culled
# lists as an order-preserving subset) on 150 seeded random scenes
mkdir -p gpurun_out/r06
( time DGS_FUZZ_SWEEP=150 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bit_exact" -p no:cacheprovider ) > gpurun_out/r06/bit_exact_sweep.log 2>&1
grep -n "^E   \|^FAILED\|passed\|failed" gpurun_out/r06/bit_exact_sweep.log | cut -c1-220 | head -40
