#!/bin/bash
# round 6 call 26: bit-identity between execution modes (tile-culled vs reference lists, rerun, wide vs packed records, K fused vs
# K single calls, capacity vs two-phase forward) on 100 seeded random scenes
mkdir -p gpurun_out/r06
( time DGS_FUZZ_SWEEP=100 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "execution_modes_agree" -p no:cacheprovider ) > gpurun_out/r06/modes_sweep.log 2>&1
grep -n "^E   \|^FAILED\|passed\|failed" gpurun_out/r06/modes_sweep.log | cut -c1-220 | head -40
