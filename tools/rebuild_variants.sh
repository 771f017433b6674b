#!/bin/bash
# Rebuilds every variants/src/<name>.<unit>.hip against the CURRENT objects (run right before a gpurun A/B).
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root && python -m deblurgs_amd.build > /dev/null
rm -f variants/*.so
for f in variants/src/*.hip; do
  b=$(basename $f .hip); name=${b%%.*}; unit=${b#*.}
  tools/build_variant.sh $name $unit.hip $f > /dev/null 2>&1 && echo "built $name ($unit)" || echo "FAILED $name"
done
