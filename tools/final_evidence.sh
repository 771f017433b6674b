#!/bin/bash
# Everything the round's DESIGN / profiles quote, produced in one GPU session into gpurun_out/profiles_<round>/:
#   profiles (tools/collect_profiles.sh), bench lines per config, the config tests' own parity report, the CU-mask
#   overlap probe (round 3), soak + list stress + p2p repeats, the RCCL smoke test.   usage: tools/final_evidence.sh [round]
rnd=${1:-r06}
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
out=$root/gpurun_out/profiles_$rnd
mkdir -p $out
tools/collect_profiles.sh $rnd > $out/collect.log 2>&1
for cfg in metric cfg2 cfg1 cfg3 cfg5; do
  extra="--no-cpu-baseline"
  steps=100
  if [ "$cfg" == "metric" ]; then extra=""; fi
  if [ "$cfg" == "cfg5" ]; then steps=20; fi
  python3 bench.py --config $cfg --steps $steps --warmup 5 $extra > $out/bench_${rnd}_$cfg.json 2> $out/bench_${rnd}_$cfg.err
done
# SURVEY 8's secondary variant: D = 3, M = 16 (bench line + kernel stats)
python3 bench.py --config metric --sh-degree 3 --steps 100 --warmup 5 --no-cpu-baseline > $out/bench_${rnd}_metric_sh3.json 2> $out/bench_${rnd}_metric_sh3.err
( cd /tmp && DGS_BWD_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $out/trace_sh3 -o trace --output-format csv -- python3 $root/bench.py --sh-degree 3 --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists --no-graph > $out/trace_sh3.log 2>&1 )
cp $(find $out/trace_sh3 -name "*kernel_stats.csv" | head -1) $out/${rnd}_kernel_stats_sh3.csv 2>/dev/null; rm -rf $out/trace_sh3
# one GPU's view of the N-GPU step: emulated shard slices -> predicted scaling table
python3 tools/predict_scaling.py --out $out/predicted_scaling_${rnd}.json > $out/predicted_scaling_${rnd}.txt 2>&1
# the metric step replayed as one hipGraph (TrainingLoop's "auto" leaves it to the eager step), the eager step with the
# backward as one launch, cfg2's step enqueued eagerly (it is replayed by default)
python3 bench.py --config metric --steps 50 --warmup 5 --no-cpu-baseline --no-reference-lists --graph-always > $out/bench_${rnd}_metric_graph_always.json 2>/dev/null
DGS_BWD_OVERLAP=0 python3 bench.py --config metric --steps 50 --warmup 5 --no-cpu-baseline --no-reference-lists --no-graph > $out/bench_${rnd}_metric_eager.json 2>/dev/null
python3 bench.py --config cfg2 --steps 50 --warmup 5 --no-cpu-baseline --no-reference-lists --no-graph > $out/bench_${rnd}_cfg2_eager.json 2>/dev/null
python3 bench.py --config metric --steps 30 --warmup 5 --no-cpu-baseline --no-reference-lists --autograd-path > $out/bench_${rnd}_metric_autograd.json 2>/dev/null
python3 tools/soak.py 2000 always > $out/soak_${rnd}.log 2>&1
python3 tools/stress_lists.py > $out/stress_lists_${rnd}.log 2>&1
python3 tools/p2p_repeat.py --repeat 10 --mode subframes > $out/${rnd}_p2p_repeat_final.log 2>&1
python3 tools/rccl_smoke.py > $out/${rnd}_rccl_smoke.txt 2>&1
python3 -m pytest tests/test_gpu_configs.py -q -s > $out/${rnd}_gpu_configs.log 2>&1
DGS_TOY_SPREAD_LIVE=1 python3 -m pytest tests/test_gpu_train.py -q -s -k "toy_deblurring or graph_replay or rccl or two_ranks or captured" > $out/${rnd}_gpu_train_extract.log 2>&1
DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 python3 bench.py --gpus 2 --config cfg2 --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_${rnd}_cfg2_2ranks_one_gpu.json 2>/dev/null
# four ranks on the one GPU (gloo): views, subframes and the 2 x 2 mesh in one line (by_mode)
DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 python3 bench.py --gpus 4 --config cfg2 --steps 30 --warmup 5 --no-cpu-baseline --extras-mesh > $out/bench_${rnd}_cfg2_4ranks_one_gpu.json 2>/dev/null
for f in $out/bench_${rnd}_*.json; do tail -1 $f | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
tail -3 $out/${rnd}_gpu_configs.log
