#!/bin/bash
# round 6 call 5: whole GPU suite on the ABI-14 tree, with durations (baseline for the "< 400 s" item)
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r06/suite_abi14.log 2>&1
tail -60 gpurun_out/r06/suite_abi14.log
