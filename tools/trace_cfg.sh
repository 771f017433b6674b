#!/bin/bash
# rocprofv3 kernel trace of bench.py for one config (graph replay on): per-kernel stats and the replayed step's timeline
# usage: tools/trace_cfg.sh <config> [extra bench args]
cfg=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
export TMPDIR=/tmp
out=$root/gpurun_out/trace_$cfg
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 bench.py --config $cfg --steps 50 --warmup 5 --no-cpu-baseline --no-reference-lists "$@" > $out/trace.log 2>&1
python3 - "$out" <<'PY'
import csv, sys, glob
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels total ms:", tot / 1e6, "distinct:", len(rows), "launches:", sum(int(r["Calls"]) for r in rows))
for r in rows[:25]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["TotalDurationNs"])/tot*100:5.1f}')
# timeline of the last complete step: gaps between consecutive kernels
t = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
k = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
k = k[-400:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in k)
span = int(k[-1]["End_Timestamp"]) - int(k[0]["Start_Timestamp"])
print(f"last 400 launches: span {span/1e3:.1f} us, busy {busy/1e3:.1f} us ({busy/span*100:.0f} %), mean gap {(span-busy)/399/1e3:.2f} us")
PY
grep "\"metric\"" $out/trace.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"config\"][\"workload\"][:40], d[\"value\"], d[\"ms_per_step\"])"
