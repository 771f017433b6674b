#!/bin/bash
# Builds variants/libdgs_<name>.so from the current objects with ONE translation unit replaced (A/B kernel experiments;
# select at run time with DGS_LIB_PATH).  usage: tools/build_variant.sh <name> <unit.hip> <source file> [extra hipcc flags]
set -e
name=$1; unit=$2; src=$3; shift 3
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/deblurgs_amd/csrc/obj
declare -A extra=( [preprocess]="-ffp-contract=off" [binning]="-ffp-contract=off" [composite]="-fno-slp-vectorize" [optim]="-ffp-contract=off" )
base=${unit%.hip}
mkdir -p $root/variants/obj
cp "$src" $root/deblurgs_amd/csrc/_variant_$base.hip
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function ${extra[$base]} "$@" \
  -c $root/deblurgs_amd/csrc/_variant_$base.hip -o $root/variants/obj/${name}_$base.o
rm -f $root/deblurgs_amd/csrc/_variant_$base.hip
objs=""
for o in preprocess binning composite geometry_bwd pose knn optim api; do
  if [ "$o" == "$base" ]; then objs="$objs $root/variants/obj/${name}_$base.o"; else objs="$objs $obj/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/variants/libdgs_$name.so $objs
echo $root/variants/libdgs_$name.so
