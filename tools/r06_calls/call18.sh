#!/bin/bash
# round 6 call 18: the three policy tests in the default mode and the two non-default ones (call 17's failures were their
# default-policy assertions, not the library)
mkdir -p gpurun_out/r06
K="graph_replay_equals_eager_fused_step or two_ranks_bench_launcher or auto_graph_policy"
for mode in "" "DGS_BWD_OVERLAP=2" "DGS_TILE_CULL=0"; do
  echo "== ${mode:-default}"
  env $mode timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -k "$K" 2>&1 | tail -3
done 2>&1 | tee gpurun_out/r06/policy_tests_modes.log
