#!/bin/bash
# round 6 call 6: new tests (L0 _C module, forward-only path, policy test), the new bench line, and where the 100 s of
# test_config_as_benchmarked[metric] go (oracle forward / unstable masks / the three backward modes / HIP side)
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -x -q -k "l0_C or forward_only or auto_graph_policy or backward_in_parts or extra_region" > gpurun_out/r06/new_tests.log 2>&1
tail -15 gpurun_out/r06/new_tests.log
timeout 900 python bench.py --steps 40 --warmup 5 > gpurun_out/r06/bench_new.json 2> gpurun_out/r06/bench_new.err
tail -c 6000 gpurun_out/r06/bench_new.json; tail -5 gpurun_out/r06/bench_new.err
timeout 900 python tools/r06_calls/oracle_timing.py > gpurun_out/r06/oracle_timing.log 2>&1
cat gpurun_out/r06/oracle_timing.log
