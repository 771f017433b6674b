#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q -s -k "as_benchmarked and metric" > $OUT/c17_metric_report.log 2>&1; echo "rc $?" >> $OUT/c17_metric_report.log; grep -i "conic\|passed\|failed\|rc " $OUT/c17_metric_report.log | tail
timeout 600 python tools/rccl_smoke.py > $OUT/c17_rccl_smoke.log 2>&1; echo "rc $?" >> $OUT/c17_rccl_smoke.log; grep "rccl smoke\|rc " $OUT/c17_rccl_smoke.log
