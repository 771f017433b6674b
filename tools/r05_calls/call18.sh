#!/bin/bash
# r05 call 18: depth order without the invisible pairs: binning tests at every size, stress lists, A/B vs the committed build
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/c18_parity.log 2>&1; echo "parity rc=$?"; tail -3 $OUT/c18_parity.log
timeout 600 python tools/stress_lists.py > $OUT/c18_stress.log 2>&1; tail -1 $OUT/c18_stress.log
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -x -k "cfg2 or metric or cfg5" > $OUT/c18_configs.log 2>&1; echo "configs rc=$?"; tail -2 $OUT/c18_configs.log
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "fused or graph or toy or overflow" > $OUT/c18_train.log 2>&1; echo "train rc=$?"; tail -2 $OUT/c18_train.log
python tools/ab_bench.py --steps 40 variants/libdgs_head.so default 2>&1 | tee $OUT/c18_ab_dropinvisible.log
