#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp; R=$PWD
for lib in head new; do
  if [ $lib = head ]; then export DGS_LIB_PATH=$R/variants/libdgs_head.so; else unset DGS_LIB_PATH; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05/ks_$lib -o t --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists --no-graph > /dev/null 2>&1 )
  echo "== $lib"; python3 tools/kernel_stats_grep.py gpurun_out/r05/ks_$lib 23 dsort cull scan gather
  rm -rf gpurun_out/r05/ks_$lib
done
