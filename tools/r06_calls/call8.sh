#!/bin/bash
mkdir -p gpurun_out/r06
L=gpurun_out/r06/oracle_par.log
: > $L
for cfg in "1 256" "1 32" "8 32" "4 64" "16 16" "8 16"; do
  timeout 300 python tools/r06_calls/oracle_par.py $cfg 2>/dev/null >> $L
done
for cfg in "8 32" "16 16"; do
  OMP_WAIT_POLICY=passive timeout 300 python tools/r06_calls/oracle_par.py $cfg 2>/dev/null >> $L
  OMP_WAIT_POLICY=passive OMP_PROC_BIND=false timeout 300 python tools/r06_calls/oracle_par.py $cfg 2>/dev/null >> $L
done
cat $L; nproc; lscpu | grep -E "NUMA|Socket|Model name" 
