#!/bin/bash
# r05 call 3: where the sharded step's extra time goes -- kernel traces of the single-GPU step, the emulated "views" slice
# and the emulated "subframes" slice (rank 1 of 8), eager, same box.
OUT=gpurun_out/r05
mkdir -p $OUT
export PYTHONPATH=$PWD TMPDIR=/tmp
R=$PWD
trace() {  # name, bench args...
  name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/tr_$name -o trace --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference-lists --no-graph "$@" > $R/$OUT/c3_$name.log 2>&1 )
  cp $(find $OUT/tr_$name -name "*kernel_stats.csv" | head -1) $OUT/c3_kernel_stats_$name.csv 2>/dev/null
  rm -rf $OUT/tr_$name
  tail -1 $OUT/c3_$name.log | cut -c1-400
}
trace single
trace views --shard views --emulate-shard 0/8
trace subframes --shard subframes --emulate-shard 1/8
python - <<'PY'
import csv, collections
def load(n):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f"gpurun_out/r05/c3_kernel_stats_{n}.csv")):
        name = r["Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        d[name] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return d
s, v, u = load("single"), load("views"), load("subframes")
steps = 23.0
print(f"{'kernel':62s} {'single ms/step':>14s} {'views':>10s} {'subframes':>10s}")
for k in sorted(set(s) | set(v) | set(u), key=lambda k: -(v.get(k, (0, 0))[1])):
    a, b, c = s.get(k, (0, 0)), v.get(k, (0, 0)), u.get(k, (0, 0))
    if max(a[1], b[1], c[1]) / steps < 0.004: continue
    print(f"{k:62s} {a[1]/steps:8.3f} x{a[0]:<5d} {b[1]/steps:6.3f} x{b[0]:<5d} {c[1]/steps:6.3f} x{c[0]:<5d}")
print("total", sum(x[1] for x in s.values())/steps, sum(x[1] for x in v.values())/steps, sum(x[1] for x in u.values())/steps)
PY
