#!/bin/bash
# part shapes for the eagerly enqueued step (the default at this size): 4 rounds interleaved, 60 steps
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call31.log
: > $L
for rep in 1 2 3 4; do
  for parts in "7,6" "6,5" "8,5" "5,5,3" "6,6,2" "9,5" "10,4" "11"; do
    echo -n "parts='$parts' " >> $L
    DGS_BWD_PARTS=$parts timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
  done
done
sort $L | cut -c1-50
