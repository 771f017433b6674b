#!/bin/bash
# r05 call 15: what is left of the sharded step's overhead: single vs emulated views slice on one box + idle gaps
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
for rep in 1 2; do
python bench.py --steps 60 --warmup 4 --no-cpu-baseline --no-reference-lists 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('single', d['ms_per_step'], d['config']['graph'])"
python bench.py --steps 60 --warmup 4 --shard views --emulate-shard 0/8 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])['emulated_shard']; print('views', d['ms_per_step'], d['graph'])"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$OUT/gap_v -o t --output-format csv -- python3 $R/bench.py --steps 60 --warmup 4 --shard views --emulate-shard 0/8 > /dev/null 2>&1 )
python3 tools/step_gaps.py $OUT/gap_v alignment_fwd_kernel first | head -24 | tee $OUT/c15_gaps_views.txt
rm -rf $OUT/gap_v
