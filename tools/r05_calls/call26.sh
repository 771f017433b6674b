#!/bin/bash
# is the device memory that the forked capture loses lost per CAPTURE or per REPLAY?  1M Gaussians, four views, no
# densification: four captures, then replays only
mkdir -p gpurun_out/r05
for m in 1 0; do
  DGS_BWD_OVERLAP=$m timeout 600 python tools/soak.py 600 always replays > gpurun_out/r05/c26_replays_overlap$m.log 2>&1
  echo "== DGS_BWD_OVERLAP=$m"; grep "^it " gpurun_out/r05/c26_replays_overlap$m.log | awk 'NR%4==1' | cut -c1-150; tail -2 gpurun_out/r05/c26_replays_overlap$m.log | cut -c1-200
done
