#!/bin/bash
# r05 call 4: persistent compositing kernels (dynamic tile tickets) -- parity suite, same-box A/B against the round-5 base
# build and the forced-6-waves variant, the timeline of the new schedule; geometry_bwd_kernel<16> with the split SH loop.
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/c4_parity_suite.log 2>&1; echo "parity suite rc=$?"; tail -3 $OUT/c4_parity_suite.log
timeout 900 python tools/ab_bench.py --steps 40 variants/libdgs_r5base.so default variants/libdgs_persist_w6.so > $OUT/c4_ab_persistent.log 2>&1
cat $OUT/c4_ab_persistent.log
DGS_LIB_PATH=$PWD/variants/libdgs_timeline.so timeout 600 python tools/tile_timeline.py --json $OUT/tile_timeline_metric_persistent.json > $OUT/c4_timeline.log 2>&1; echo "timeline rc=$?"
tail -32 $OUT/c4_timeline.log
for d in 3 0; do
timeout 900 python tools/ab_bench.py --steps 30 --extra "--sh-degree $d" variants/libdgs_r5base.so default > $OUT/c4_ab_sh$d.log 2>&1; cat $OUT/c4_ab_sh$d.log
done
timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -k "cfg2" > $OUT/c4_cfg2.log 2>&1; echo "cfg2 (+sh3) rc=$?"; tail -2 $OUT/c4_cfg2.log
bash tools/r05_calls/call3.sh
