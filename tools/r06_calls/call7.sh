#!/bin/bash
# round 6 call 7: the new tests again, the with-depth anomaly, relu mask in the row word + conditional totals load (A/B of
# the geometry stage against call 6's 0.648 / preprocess 0.272), full-size oracle checks with the oracle's new threading
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -x -q -k "l0_C or forward_only or preprocess_bit_exact or backward_vs_oracle or variants or fused_equals or raw_param or cloud" > gpurun_out/r06/new_tests2.log 2>&1
tail -8 gpurun_out/r06/new_tests2.log
timeout 600 python tools/r06_calls/depth_debug.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/depth_debug.log
for i in 1 2; do timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py; done | tee gpurun_out/r06/bench_relumask.log
( time timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "metric or cfg2" ) > gpurun_out/r06/configs_threads.log 2>&1
tail -12 gpurun_out/r06/configs_threads.log
