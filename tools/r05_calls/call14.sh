#!/bin/bash
# r05 call 14: the sharded step without pageable host-to-device copies (A/B of emulated slices against the previous tree)
OUT=gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for rep in 1 2; do
  for tree in prev new; do
    if [ $tree = prev ]; then d=$R/variants/src/prev; else d=$R; fi
    for mode in "views 0/8" "subframes 1/8"; do
      set -- $mode
      ( cd $d && PYTHONPATH=$d python bench.py --steps 60 --warmup 4 --shard $1 --emulate-shard $2 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])['emulated_shard']; print('$tree', '$1', d['ms_per_step'], d.get('eager_ms_per_step'), d['graph'], json.dumps(d.get('rccl'))[:900])" )
    done
  done
done 2>&1 | tee $OUT/c14_ab_sharded.log
export PYTHONPATH=$R
python bench.py --config cfg2 --steps 10 --warmup 2 --shard views --emulate-shard 0/8 2>/dev/null | tail -3 | cut -c1-300
