#!/bin/bash
# eagerly enqueued fused step with the backward in parts against the graph replay (no fork inside a capture): more samples,
# and the emulated "views" slice of an 8-GPU step both ways
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call29.log
: > $L
for rep in 1 2 3 4 5 6; do
  for g in "" "--no-graph"; do
    echo -n "metric $g " >> $L
    timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-reference-lists $g 2>/dev/null | python tools/r05_calls/brief.py >> $L
  done
done
for rep in 1 2 3; do
  for g in "" "--no-graph"; do
    echo -n "views-slice $g " >> $L
    timeout 600 python bench.py --emulate-shard 0/8 --shard views --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists $g 2>/dev/null | python tools/r05_calls/brief.py >> $L
    echo -n "sh3 $g " >> $L
    timeout 600 python bench.py --sh-degree 3 --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists $g 2>/dev/null | python tools/r05_calls/brief.py >> $L
  done
done
sort $L | cut -c1-110
