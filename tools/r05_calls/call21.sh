#!/bin/bash
# call20's mode 3 generalised to n parts (DGS_BWD_PARTS): bit-identity against the single launch + A/B of part shapes
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call21.log
: > $L
for cfg in cfg2 metric; do
  DGS_BWD_OVERLAP=0 python tools/grad_hash.py $cfg > gpurun_out/r05/hash_${cfg}_0.txt 2>&1
  for parts in "" "5,5" "4,4,4" "1" "14"; do
    DGS_BWD_OVERLAP=3 DGS_BWD_PARTS=$parts python tools/grad_hash.py $cfg > gpurun_out/r05/hash_${cfg}_p.txt 2>&1
    if cmp -s gpurun_out/r05/hash_${cfg}_0.txt gpurun_out/r05/hash_${cfg}_p.txt; then echo "$cfg parts='$parts': bit-identical" >> $L; else echo "$cfg parts='$parts': DIFFERENT" >> $L; diff gpurun_out/r05/hash_${cfg}_0.txt gpurun_out/r05/hash_${cfg}_p.txt | head -8 >> $L; fi
  done
done
cat gpurun_out/r05/hash_metric_0.txt >> $L
DGS_BWD_OVERLAP=3 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward_vs_oracle or backward_is_deterministic or tile_cull_gradients_bitwise or fused_equals_per_subframe or variants" >> $L 2>&1
DGS_BWD_OVERLAP=3 timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "graph_replay_equals_eager or fused_step_equals_autograd or two_ranks_subframes or captured_front" >> $L 2>&1
for rep in 1 2; do
  for parts in off "" "8,4" "5,5" "6,5" "4,4,4" "3,3,3,3" "5,4,3,2" "2,2,2,2,2,2,2"; do
    echo "== bench parts='$parts'" >> $L
    if [ "$parts" = off ]; then
      DGS_BWD_OVERLAP=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
    else
      DGS_BWD_OVERLAP=3 DGS_BWD_PARTS=$parts timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
    fi
  done
done
tail -80 $L
