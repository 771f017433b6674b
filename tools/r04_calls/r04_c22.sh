#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/c22_smoke.log 2>&1; tail -2 $OUT/c22_smoke.log
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -k "captured_front or bench_launcher or replicas" > $OUT/c22_tests.log 2>&1; tail -2 $OUT/c22_tests.log
python bench.py --steps 20 --warmup 3 > $OUT/c22_bench.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/c22_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('valu',{}).get('frac'), d['cpu_baseline']['value'])"
