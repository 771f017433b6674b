#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests/test_gpu_train.py -x -q > $OUT/c8_gpu_train.log 2>&1; echo "rc $?" >> $OUT/c8_gpu_train.log; tail -15 $OUT/c8_gpu_train.log
DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --config cfg2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c8_bench2_cfg2.json 2> $OUT/c8_bench2_cfg2.err; tail -c 1500 $OUT/c8_bench2_cfg2.json
DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --config cfg2 --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $OUT/c8_bench2_cfg2_eager.json 2> $OUT/c8_bench2_cfg2_eager.err; python -c "
import json
for f in ('c8_bench2_cfg2.json','c8_bench2_cfg2_eager.json'):
    d=json.loads(open('$OUT/'+f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['config']['graph'])
"
