#!/bin/bash
# round 6 call 2: EXEC-half microbenchmark + kernel stats of the round-5 tree on this round's box
mkdir -p gpurun_out/r06
tools/exec_half > gpurun_out/r06/exec_half.txt 2>&1
cat gpurun_out/r06/exec_half.txt
export TMPDIR=/tmp
DGS_BWD_OVERLAP=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r06/prof_base -o base --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists --no-graph > /dev/null 2>&1
python tools/kernel_stats_grep.py gpurun_out/r06/prof_base 23 "" 2>/dev/null | sort -k7 -n -r | head -40
