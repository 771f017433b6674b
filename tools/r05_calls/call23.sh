#!/bin/bash
# the backward in parts as the library's default: the new test, the tests that run at sizes where it is on, A/B on/off
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call23.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "backward_in_parts or capacity_mode or graph_replay or overflow or two_ranks_subframes or emulated" >> $L 2>&1
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -m gpu -x >> $L 2>&1
for rep in 1 2 3; do
  for m in 0 1; do
    echo -n "DGS_BWD_OVERLAP=$m " >> $L
    DGS_BWD_OVERLAP=$m timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
    echo -n "DGS_BWD_OVERLAP=$m sh3 " >> $L
    DGS_BWD_OVERLAP=$m timeout 600 python bench.py --sh-degree 3 --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
    echo -n "DGS_BWD_OVERLAP=$m cfg3 " >> $L
    DGS_BWD_OVERLAP=$m timeout 600 python bench.py --config cfg3 --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python tools/r05_calls/brief.py >> $L
  done
done
grep -v "^$" $L | tail -40 | cut -c1-150
