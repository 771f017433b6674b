#!/bin/bash
# Collects the round's committed evidence on the GPU box into gpurun_out/profiles_<round>/ (copy what is to be judged into
# profiles/ afterwards): kernel-trace summary, VALU counters, FETCH / WRITE counters (separate passes, kernel-trace only),
# the per-class issue costs, and the derived json files.   usage: tools/collect_profiles.sh [round]
rnd=${1:-r06}
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
export TMPDIR=/tmp
out=$root/gpurun_out/profiles_$rnd
mkdir -p $out
# every profiler pass runs the compositing backward as ONE launch per step (DGS_BWD_OVERLAP=0): the launch the bench line's
# `roofline` is about; one more trace is taken with the library's default (eager step, the backward in parts, each part's row
# totals next to the next part's compositing: kernel durations then overlap and do not add up to the step)
export DGS_BWD_OVERLAP=0
B="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists --no-graph"
tools/valu_rate > $out/valu_classes_$rnd.txt 2>&1
cp $out/valu_classes_$rnd.txt profiles/valu_classes_$rnd.txt
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -d $out/valu_peak -o pmc --output-format csv -- $root/tools/valu_rate 8 > $out/valu_peak.log 2>&1 )
python3 profiles/make_valu_peak.py $out/valu_peak $out/valu_classes_$rnd.txt > profiles/valu_peak_$rnd.json 2> $out/valu_peak_err.log
python3 tools/isa_census.py --json profiles/isa_census_$rnd.json > $out/isa_census.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists --no-graph > $out/trace.log 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/${rnd}_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $out/trace_graph -o trace --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists --graph-always > $out/trace_graph.log 2>&1
cp $(find $out/trace_graph -name "*kernel_stats.csv" | head -1) $out/${rnd}_kernel_stats_graph_replay.csv
# the product's default at this size: the eagerly enqueued step with the backward in parts (kernel durations overlap)
DGS_BWD_OVERLAP=1 rocprofv3 --kernel-trace --stats -d $out/trace_parts -o trace --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists > $out/trace_parts.log 2>&1
cp $(find $out/trace_parts -name "*kernel_stats.csv" | head -1) $out/${rnd}_kernel_stats_default_parts.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $out/valu -o pmc --output-format csv -- $B > $out/valu.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o pmc --output-format csv -- $B > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o pmc --output-format csv -- $B > $out/write.log 2>&1
R=$(grep -o '"R_total": [0-9]*' $out/valu.log | head -1 | grep -o '[0-9]*$')
python3 profiles/make_valu.py $out/valu metric $rnd $R > $out/make_valu.log 2>&1
python3 profiles/make_traffic.py $out/fetch $out/write metric $rnd 2 > $out/make_traffic.log 2>&1
cp profiles/valu_$rnd.json profiles/traffic_$rnd.json profiles/valu_peak_$rnd.json profiles/isa_census_$rnd.json $out/ 2>/dev/null
unset DGS_BWD_OVERLAP
tail -3 $out/make_valu.log; echo "R_total=$R"; ls $out
