#!/bin/bash
# the long-first cut (K = 15: 12, 2, 1) as the library's default against the previous 7, 6, 2 at every large configuration;
# then the tests that run the parts at full size, and the bench lines / parts trace / predicted scaling of the final build
mkdir -p gpurun_out/r05 gpurun_out/profiles_r05
export PYTHONPATH=$PWD TMPDIR=/tmp
L=gpurun_out/r05/call33.log
O=gpurun_out/profiles_r05
: > $L
for rep in 1 2 3; do
  for parts in default "7,6"; do
    for cfg in "--config metric" "--config metric --sh-degree 3" "--config cfg3"; do
      echo -n "parts=$parts $cfg: " >> $L
      if [ "$parts" = default ]; then timeout 600 python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
      else DGS_BWD_PARTS=$parts timeout 600 python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L; fi
    done
  done
done
for parts in default "14,13" default "14,13"; do
  echo -n "parts=$parts cfg5: " >> $L
  if [ "$parts" = default ]; then timeout 900 python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
  else DGS_BWD_PARTS=$parts timeout 900 python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L; fi
done
sort $L | cut -c1-90
python3 -m pytest tests/test_gpu_configs.py -q -s > $O/r05_gpu_configs.log 2>&1; tail -2 $O/r05_gpu_configs.log
python3 -m pytest tests/test_gpu_train.py -q -m gpu -k "parts or policy or capacity_mode or toy_deblurring" 2>&1 | tail -2
for cfg in metric cfg3 cfg5; do python3 tools/grad_hash.py $cfg > /tmp/h1_$cfg.txt 2>/dev/null; DGS_BWD_OVERLAP=0 python3 tools/grad_hash.py $cfg > /tmp/h0_$cfg.txt 2>/dev/null; if cmp -s /tmp/h0_$cfg.txt /tmp/h1_$cfg.txt; then echo "$cfg: default cut bit-identical to the single launch" | tee -a $L; else echo "$cfg: DIFFERENT" | tee -a $L; fi; done
python3 bench.py > $O/bench_r05_metric.json 2> $O/bench_r05_metric.err; echo "bench rc=$?"
python3 bench.py --sh-degree 3 --no-cpu-baseline > $O/bench_r05_metric_sh3.json 2> /dev/null
for cfg in cfg3 cfg5; do python3 bench.py --config $cfg --no-cpu-baseline > $O/bench_r05_$cfg.json 2>/dev/null; done
python3 bench.py --config metric --steps 30 --warmup 5 --no-cpu-baseline --no-reference-lists --autograd-path > $O/bench_r05_metric_autograd.json 2>/dev/null
python3 tools/predict_scaling.py --out $O/predicted_scaling_r05.json > $O/predicted_scaling_r05.txt 2>&1
DGS_BWD_OVERLAP=1 rocprofv3 --kernel-trace --stats -d $O/trace_parts -o trace --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists > $O/trace_parts.log 2>&1
cp $(find $O/trace_parts -name "*kernel_stats.csv" | head -1) $O/r05_kernel_stats_default_parts.csv
for f in $O/bench_r05_metric.json $O/bench_r05_metric_sh3.json $O/bench_r05_cfg3.json $O/bench_r05_cfg5.json $O/bench_r05_metric_autograd.json; do grep '^{' $f | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
tail -8 $O/predicted_scaling_r05.txt
