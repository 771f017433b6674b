#!/bin/bash
# r05 call 12: surviving-tile counts carried through the depth sort (no gather): binning tests at every size, A/B vs prev tree
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/c12_parity.log 2>&1; echo "parity rc=$?"; tail -2 $OUT/c12_parity.log
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -x -k "cfg2 or metric or cfg5" > $OUT/c12_configs.log 2>&1; echo "configs rc=$?"; tail -2 $OUT/c12_configs.log
timeout 600 python tools/stress_lists.py > $OUT/c12_stress.log 2>&1; tail -1 $OUT/c12_stress.log
for rep in 1 2; do
  for tree in prev new; do
    if [ $tree = prev ]; then d=$R/variants/src/prev; else d=$R; fi
    ( cd $d && PYTHONPATH=$d python bench.py --config metric --steps 100 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['stages']; print('$tree', d['value'], d['ms_per_step'], {k: st[k]['avg_ms'] for k in ('depth_order','tile_cull','scan','duplicate','sort') if k in st})" )
  done
done 2>&1 | tee $OUT/c12_ab_counts.log
