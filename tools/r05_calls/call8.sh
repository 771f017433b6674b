#!/bin/bash
# r05 call 8: the whole GPU suite (timing + durations), predicted scaling with the per-step geometry time
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu --durations=12 > $OUT/c8_gpu_suite.log 2>&1; echo "suite rc=$?"
tail -22 $OUT/c8_gpu_suite.log
timeout 1500 python tools/predict_scaling.py --out $OUT/predicted_scaling.json > $OUT/c8_predict.log 2>&1; echo "predict rc=$?"
tail -9 $OUT/c8_predict.log
