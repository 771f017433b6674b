#!/bin/bash
# r05 call 1: new tests (ADVICE r4 fixes, driver launch form, emulated slices, pose golden, cfg1, SH3), same-box baseline
# bench, the D=3 / M=16 secondary bench + kernel stats, predicted scaling table.
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "overflow or refused or drivers_launch or emulated" > $OUT/c1_tests_train.log 2>&1; echo "train tests rc=$?" 
tail -5 $OUT/c1_tests_train.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "pose_kernel_on or cfg1_exact" > $OUT/c1_tests_parity.log 2>&1; echo "parity tests rc=$?"
tail -4 $OUT/c1_tests_parity.log
timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -s -k "cfg2_sh3" > $OUT/c1_tests_sh3.log 2>&1; echo "sh3 test rc=$?"
tail -12 $OUT/c1_tests_sh3.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/c1_bench_metric.json 2> $OUT/c1_bench_metric.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05/c1_bench_metric.json").read().strip().splitlines()[-1])
print("metric:", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["stages"].items()})
PY
timeout 600 python bench.py --sh-degree 3 --no-cpu-baseline > $OUT/c1_bench_metric_sh3.json 2> $OUT/c1_bench_metric_sh3.err; echo "bench sh3 rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05/c1_bench_metric_sh3.json").read().strip().splitlines()[-1])
print("sh3:", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["stages"].items()})
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/trace_sh3 -o trace --output-format csv -- python3 $OLDPWD/bench.py --sh-degree 3 --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists --no-graph > $OLDPWD/$OUT/c1_trace_sh3.log 2>&1 )
cp $(find $OUT/trace_sh3 -name "*kernel_stats.csv" | head -1) $OUT/r05_kernel_stats_sh3.csv 2>/dev/null
head -12 $OUT/r05_kernel_stats_sh3.csv
rm -rf $OUT/trace_sh3
timeout 1500 python tools/predict_scaling.py --out $OUT/predicted_scaling.json > $OUT/c1_predict.log 2>&1; echo "predict rc=$?"
cat $OUT/c1_predict.log | tail -12
