#!/bin/bash
# round 6 call 15: whole GPU suite with durations
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -q -s --durations=30 > gpurun_out/r06/suite_3.log 2>&1
grep -E "^\[(metric|cfg3|cfg5)\] [0-9]+ s|passed|failed|^[0-9.]+s call|^FAILED|^E  " gpurun_out/r06/suite_3.log | cut -c1-300 | tail -60
