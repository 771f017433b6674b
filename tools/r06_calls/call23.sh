#!/bin/bash
# round 6 call 23: the fuzz test on 120 more seeded random cases (DGS_FUZZ_SWEEP): a wider net than the nine of the suite
mkdir -p gpurun_out/r06
( time DGS_FUZZ_SWEEP=120 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz_shapes -p no:cacheprovider ) > gpurun_out/r06/fuzz_sweep.log 2>&1
tail -30 gpurun_out/r06/fuzz_sweep.log | cut -c1-260
