#!/bin/bash
# VALU issue-rate evidence: the micro-benchmark by itself (all occupancies), then under rocprofv3 PMC at 8 waves per SIMD.
OUT=gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
$R/tools/valu_rate > $R/$OUT/valu_classes_r04.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -d $R/$OUT/valu_peak -o pmc --output-format csv -- $R/tools/valu_rate 8 > $R/$OUT/valu_peak.log 2>&1
cd $R
python3 profiles/make_valu_peak.py $OUT/valu_peak $OUT/valu_classes_r04.txt > $OUT/valu_peak_r04.json 2> $OUT/valu_peak_err.log
cat $OUT/valu_classes_r04.txt | grep "waves/SIMD 8"; cat $OUT/valu_peak_r04.json
