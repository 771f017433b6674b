#!/bin/bash
# r05 call 10: GPU timeline of the sharded ("views") step against the single-GPU step: idle gaps per step
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
for name in single views; do
  extra=""; if [ $name = views ]; then extra="--shard views --emulate-shard 0/8"; fi
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$OUT/gap_$name -o t --output-format csv -- python3 $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-reference-lists $extra > $R/$OUT/c10_$name.log 2>&1 )
  echo "== $name"; python3 tools/step_gaps.py $OUT/gap_$name | tee $OUT/c10_gaps_$name.txt
  rm -rf $OUT/gap_$name
done
