#!/bin/bash
# soak (graph = always: a re-capture of every view after every densification) with and without the backward in parts:
# device memory in use next to torch's own figures -- the refreshed soak of the evidence run died out of device memory
mkdir -p gpurun_out/r05
for m in 1 0; do
  DGS_BWD_OVERLAP=$m timeout 600 python tools/soak.py 700 always > gpurun_out/r05/c25_soak_overlap$m.log 2>&1
  echo "== DGS_BWD_OVERLAP=$m"; grep "^it " gpurun_out/r05/c25_soak_overlap$m.log | awk 'NR%4==0' | cut -c1-150; tail -2 gpurun_out/r05/c25_soak_overlap$m.log | cut -c1-200
done
