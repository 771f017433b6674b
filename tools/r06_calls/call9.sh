#!/bin/bash
# round 6 call 9: whole GPU suite with durations after the oracle threading / test restructuring
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=25 > gpurun_out/r06/suite_2.log 2>&1
grep -E "^\[(metric|cfg3|cfg5|cfg2)\] [0-9]+ s|passed|failed|^[0-9.]+s call" gpurun_out/r06/suite_2.log | cut -c1-400 | tail -45
