#!/bin/bash
# the default bench line alone (the box of the last final_check.sh call ran 3 % below the others)
mkdir -p gpurun_out/profiles_r06
python3 bench.py --steps 100 --warmup 5 > gpurun_out/profiles_r06/bench_r06_metric.json 2> gpurun_out/profiles_r06/bench_r06_metric.err
tail -1 gpurun_out/profiles_r06/bench_r06_metric.json | python tools/brief.py
