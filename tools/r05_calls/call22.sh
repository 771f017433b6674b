#!/bin/bash
# repeated A/B of the part shapes (graph replay, 40 steps, 4 rounds interleaved)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call22.log
: > $L
for rep in 1 2 3 4; do
  for parts in off "6,5" "4,4,4" "7,6" "6,6" "5,5,3" "9,4"; do
    echo -n "parts='$parts' " >> $L
    if [ "$parts" = off ]; then
      DGS_BWD_OVERLAP=0 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
    else
      DGS_BWD_OVERLAP=3 DGS_BWD_PARTS=$parts timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
    fi
  done
done
sort $L | cut -c1-60
