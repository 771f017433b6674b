#!/bin/bash
# Builds variants/libdgs_<name>.so = the whole library compiled with extra hipcc flags (e.g. -DDGS_SUMS_F=16): A/B
# experiments that span several translation units.  usage: tools/build_flag_variant.sh <name> <flags...>
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/variants/obj_$name
declare -A extra=( [preprocess]="-ffp-contract=off" [binning]="-ffp-contract=off" [composite]="-fno-slp-vectorize" [optim]="-ffp-contract=off" )
objs=""
for o in preprocess binning composite geometry_bwd pose knn optim api; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function \
    ${extra[$o]} "$@" -c $root/deblurgs_amd/csrc/$o.hip -o $root/variants/obj_$name/$o.o &
  objs="$objs $root/variants/obj_$name/$o.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/variants/libdgs_$name.so $objs
echo $root/variants/libdgs_$name.so
