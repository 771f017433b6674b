#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=5 > $OUT/gpu_suite_d.log 2>&1; echo "rc $?" >> $OUT/gpu_suite_d.log; tail -4 $OUT/gpu_suite_d.log
python tools/rccl_smoke.py > $OUT/r04_rccl_smoke.txt 2>&1; tail -2 $OUT/r04_rccl_smoke.txt
python bench.py > $OUT/bench_final.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['graph'])"
