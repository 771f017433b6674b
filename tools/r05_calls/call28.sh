#!/bin/bash
# what the parts are worth where they still run: eagerly enqueued steps (fused step without the graph; operator API +
# autograd), timed without the stage timers; graph replay for reference
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call28.log
: > $L
for rep in 1 2 3; do
  for g in "--no-graph" "--autograd-path" ""; do
    for m in 0 1; do
      echo -n "DGS_BWD_OVERLAP=$m $g " >> $L
      DGS_BWD_OVERLAP=$m timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists $g 2>/dev/null | python tools/r05_calls/brief.py >> $L
    done
  done
done
sort $L | cut -c1-120
