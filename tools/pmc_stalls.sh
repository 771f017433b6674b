#!/bin/bash
# Issue / stall counters of the compositing kernels (several rocprofv3 --pmc passes, kernel-trace only).
# usage: tools/pmc_stalls.sh <tag> [lib.so]
tag=$1; lib=$2
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
export TMPDIR=/tmp
if [ -n "$lib" ] && [ "$lib" != "default" ]; then export DGS_LIB_PATH=$root/$lib; fi
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --list-avail > $out/avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_IFETCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists --no-graph > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 profiles/pmc_summary.py $out | grep -i "composite"
