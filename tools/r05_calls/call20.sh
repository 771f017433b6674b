#!/bin/bash
# compositing backward in two halves of the subframes, the first half's row totals next to the second half's compositing
# (DGS_BWD_OVERLAP = 0 off, 1 side stream least urgent, 2 most urgent, 3 totals on the side stream): bit-identity + A/B
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call20.log
: > $L
for m in 1 3; do
  echo "== tests, DGS_BWD_OVERLAP=$m" >> $L
  DGS_BWD_OVERLAP=$m timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward_vs_oracle or backward_is_deterministic or tile_cull_gradients_bitwise or fused_equals_per_subframe" >> $L 2>&1
  DGS_BWD_OVERLAP=$m timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "graph_replay_equals_eager or fused_step_equals_autograd" >> $L 2>&1
done
for rep in 1 2; do
  for m in 0 1 2 3; do
    for g in "" "--no-graph"; do
      echo "== bench DGS_BWD_OVERLAP=$m $g" >> $L
      DGS_BWD_OVERLAP=$m timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-reference-lists $g 2>/dev/null | python tools/r05_calls/brief.py >> $L
    done
  done
done
tail -60 $L
