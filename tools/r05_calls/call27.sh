#!/bin/bash
# no fork inside a stream capture (the default now): soak memory flat again?  what the parts are still worth for eagerly
# enqueued steps; the forked capture (=3) kept measurable
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call27.log
: > $L
timeout 600 python tools/soak.py 700 always > gpurun_out/r05/c27_soak_default.log 2>&1
echo "== soak, default" >> $L; grep "^it " gpurun_out/r05/c27_soak_default.log | awk 'NR%4==0' | cut -c1-150 >> $L; tail -1 gpurun_out/r05/c27_soak_default.log >> $L
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "backward_in_parts" >> $L 2>&1
for rep in 1 2 3; do
  for m in 0 1 3; do
    for g in "" "--no-graph" "--autograd-path"; do
      echo -n "DGS_BWD_OVERLAP=$m $g " >> $L
      DGS_BWD_OVERLAP=$m timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists $g 2>/dev/null | python tools/r05_calls/brief.py >> $L
    done
  done
done
cat $L | cut -c1-170
