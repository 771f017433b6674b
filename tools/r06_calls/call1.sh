#!/bin/bash
# round 6 call 1: EXEC-half microbenchmark + same-box baseline of the round-5 tree
mkdir -p gpurun_out/r06
tools/exec_half > gpurun_out/r06/exec_half.txt 2>&1
cat gpurun_out/r06/exec_half.txt
for i in 1 2; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06/bench_base_$i.json
  python tools/r05_calls/brief.py < gpurun_out/r06/bench_base_$i.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06/prof_base -o base -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/kernel_stats_grep.py gpurun_out/r06/prof_base 23 "" 2>/dev/null | sort -k7 -n -r | head -40
