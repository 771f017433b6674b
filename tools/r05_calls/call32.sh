#!/bin/bash
# part shapes for the eagerly enqueued step, second sweep: a long first part
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call32.log
: > $L
for rep in 1 2 3 4; do
  for parts in "10,4" "11,3" "12,2" "10,3" "12" "13" "14" "9,4" "7,6"; do
    echo -n "parts='$parts' " >> $L
    DGS_BWD_PARTS=$parts timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
  done
done
sort $L | cut -c1-50
