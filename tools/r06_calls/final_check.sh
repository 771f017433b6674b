#!/bin/bash
# the round's last call: what the driver does at round end, on the committed build -- the whole GPU suite, smoke(), the
# default bench line (and the driver's 20-step form)
mkdir -p gpurun_out/r06 gpurun_out/profiles_r06
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r06/suite_final.log 2>&1
tail -18 gpurun_out/r06/suite_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --steps 100 --warmup 5 > gpurun_out/profiles_r06/bench_r06_metric.json 2> gpurun_out/profiles_r06/bench_r06_metric.err
tail -1 gpurun_out/profiles_r06/bench_r06_metric.json | python tools/brief.py
( time python3 bench.py --steps 20 --warmup 3 ) 2>&1 | tail -4 | cut -c1-200
