#!/bin/bash
# round 6 call 17: the GPU suite in the two non-default modes -- the backward in parts forced on for every view
# (DGS_BWD_OVERLAP=2), and the reference's duplicate lists for every call of the package (DGS_TILE_CULL=0)
mkdir -p gpurun_out/r06
( time DGS_BWD_OVERLAP=2 timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r06/suite_parts_forced.log 2>&1
tail -6 gpurun_out/r06/suite_parts_forced.log
( time DGS_TILE_CULL=0 timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r06/suite_reference_lists.log 2>&1
tail -6 gpurun_out/r06/suite_reference_lists.log
