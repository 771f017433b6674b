#!/bin/bash
# rocprofv3 kernel-trace summary (and optionally VALU counters) of bench.py for one library build.
# usage: tools/prof_variant.sh <tag> <lib.so|default> [pmc]
tag=$1; lib=$2; mode=$3
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
export TMPDIR=/tmp
if [ "$lib" != "default" ]; then export DGS_LIB_PATH=$root/$lib; fi
out=$root/gpurun_out/prof_$tag
mkdir -p $out
if [ "$mode" == "pmc" ]; then
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $out/pmc -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists > $out/pmc.log 2>&1
else
  rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists $DGS_BENCH_EXTRA > $out/trace.log 2>&1
  f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f}')
PY
fi
