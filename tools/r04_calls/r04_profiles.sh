#!/bin/bash
# Round-4 evidence: profiles (kernel stats, VALU / LDS / traffic counters), bench lines of every config, soak.
OUT=gpurun_out/r04
mkdir -p $OUT
export PYTHONPATH=$PWD
bash tools/collect_profiles.sh r04 > $OUT/collect.log 2>&1; tail -5 $OUT/collect.log
python bench.py > $OUT/bench_r04_metric.json 2> $OUT/bench_r04_metric.err; tail -c 600 $OUT/bench_r04_metric.json
for cfg in cfg1 cfg2 cfg3; do python bench.py --config $cfg --no-cpu-baseline > $OUT/bench_r04_$cfg.json 2> $OUT/bench_r04_$cfg.err; done
python bench.py --config cfg2 --no-cpu-baseline --no-graph --no-reference-lists > $OUT/bench_r04_cfg2_eager.json 2>/dev/null
python bench.py --no-cpu-baseline --no-graph --no-reference-lists > $OUT/bench_r04_metric_eager.json 2>/dev/null
python bench.py --no-cpu-baseline --autograd-path --no-reference-lists > $OUT/bench_r04_metric_autograd.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$OUT/bench_r04_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('value_reference_lists',{}) and d['value_reference_lists'].get('value'))
    except Exception as e: print(f, 'ERR', e)
"
