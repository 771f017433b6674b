#!/bin/bash
# the backward in parts at cfg5 (K * P > 2^24: the depth sort without the count payload; 380 M duplicates, parts 14 / 13 / 4)
# and at cfg3: every output bit-identical to the single launch?
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call24.log
: > $L
for cfg in cfg5 cfg3; do
  DGS_BWD_OVERLAP=0 timeout 900 python tools/grad_hash.py $cfg > gpurun_out/r05/hash_${cfg}_0.txt 2>&1
  DGS_BWD_OVERLAP=1 timeout 900 python tools/grad_hash.py $cfg > gpurun_out/r05/hash_${cfg}_1.txt 2>&1
  if cmp -s gpurun_out/r05/hash_${cfg}_0.txt gpurun_out/r05/hash_${cfg}_1.txt; then echo "$cfg: bit-identical" >> $L; else echo "$cfg: DIFFERENT" >> $L; diff gpurun_out/r05/hash_${cfg}_0.txt gpurun_out/r05/hash_${cfg}_1.txt | head >> $L; fi
  grep -v amdgpu.ids gpurun_out/r05/hash_${cfg}_1.txt >> $L
done
cat $L
