#!/bin/bash
# r05 call 11: no memset / copy nodes in the replayed step (kernels instead; status words written to pinned host memory by the
# finalize kernel; scalars pushed by dgs_copy_words): tests, then A/B against the previous tree (variants/src/prev)
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "fused or graph or overflow or capacity or drop or makes_up or captured or refused or rccl" > $OUT/c11_train.log 2>&1; echo "train subset rc=$?"; tail -3 $OUT/c11_train.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "capacity or blur_loss or slice_loss" 2>&1 | tail -2
for cfg in metric cfg2 cfg1; do
  for rep in 1 2; do
    for tree in prev new; do
      if [ $tree = prev ]; then d=$R/variants/src/prev; else d=$R; fi
      ( cd $d && PYTHONPATH=$d python bench.py --config $cfg --steps 100 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$tree', d['value'], d['ms_per_step'], d['config']['graph'])" )
    done
  done
done 2>&1 | tee $OUT/c11_ab_nodes.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$OUT/gap_new -o t --output-format csv -- python3 $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-reference-lists > $R/$OUT/c11_trace.log 2>&1 )
python3 tools/step_gaps.py $OUT/gap_new | head -14 | tee $OUT/c11_gaps_single.txt
rm -rf $OUT/gap_new
