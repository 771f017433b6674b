#!/bin/bash
# r05 call 2: parity A/B (shipped compositing arithmetic vs the reference's letter, -DDGS_EXACT_POWER=1), its cost, and the
# wave-per-tile schedule's tail (timeline build).
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
for lib in default exactpower; do
  if [ $lib != default ]; then export DGS_LIB_PATH=$PWD/variants/libdgs_$lib.so; else unset DGS_LIB_PATH; fi
  timeout 1500 python -m pytest tests/test_gpu_configs.py -q -s -k "test_config_as_benchmarked and (cfg2 or metric) and not sh3" > $OUT/c2_parity_$lib.log 2>&1; echo "parity $lib rc=$?"
  grep -A14 "^\[cfg2\|^\[metric\|passed\|failed" $OUT/c2_parity_$lib.log | head -60
done
export DGS_LIB_PATH=$PWD/variants/libdgs_exactpower.so
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -s -k "cfg5" > $OUT/c2_parity_cfg5_exactpower.log 2>&1; echo "cfg5 exact rc=$?"
grep -A14 "^\[cfg5\|passed\|failed\|exception" $OUT/c2_parity_cfg5_exactpower.log | head -40
unset DGS_LIB_PATH
timeout 900 python tools/ab_bench.py --steps 40 default variants/libdgs_exactpower.so > $OUT/c2_ab_exactpower.log 2>&1
cat $OUT/c2_ab_exactpower.log
DGS_LIB_PATH=$PWD/variants/libdgs_timeline.so timeout 600 python tools/tile_timeline.py --json $OUT/tile_timeline_metric.json > $OUT/c2_timeline.log 2>&1; echo "timeline rc=$?"
tail -30 $OUT/c2_timeline.log
