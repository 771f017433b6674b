#!/bin/bash
# r05 final refresh (the build that is committed): GPU suite, headline + secondary bench lines, predicted scaling, RCCL smoke
O=gpurun_out/profiles_r05
mkdir -p $O
export PYTHONPATH=$PWD TMPDIR=/tmp
python3 -m pytest tests/ -q -m gpu --durations=8 > $O/r05_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/r05_gpu_suite.log
python3 bench.py > $O/bench_r05_metric.json 2> $O/bench_r05_metric.err; echo "bench rc=$?"
python3 bench.py --sh-degree 3 --no-cpu-baseline > $O/bench_r05_metric_sh3.json 2> /dev/null
for cfg in cfg1 cfg2 cfg3; do python3 bench.py --config $cfg --no-cpu-baseline > $O/bench_r05_$cfg.json 2>/dev/null; done
python3 tools/predict_scaling.py --out $O/predicted_scaling_r05.json > $O/predicted_scaling_r05.txt 2>&1
python3 tools/rccl_smoke.py > $O/r05_rccl_smoke.txt 2>&1; tail -1 $O/r05_rccl_smoke.txt
DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 python3 bench.py --gpus 2 --config cfg2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_r05_cfg2_2ranks_one_gpu.json 2>/dev/null
for f in $O/bench_r05_metric.json $O/bench_r05_metric_sh3.json $O/bench_r05_cfg1.json $O/bench_r05_cfg2.json $O/bench_r05_cfg3.json; do grep '^{' $f | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
tail -8 $O/predicted_scaling_r05.txt
