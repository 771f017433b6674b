#!/bin/bash
# graph = "auto" leaves large views to the eager fused step (backward in parts); small views are replayed: bench lines of
# every config under the product's policy, against --graph-always; the emulated 8-GPU "views" slice both ways; tests
mkdir -p gpurun_out/r05
L=gpurun_out/r05/call30.log
: > $L
cat > /tmp/brief2.py <<'P'
import json, sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln)
        e = d.get("emulated_shard", d)
        c = e.get("config", d.get("config", {}))
        print(f"{e.get('ms_per_step')} ms/step  graph {c.get('graph') if isinstance(c, dict) else None}  eager_preferred {c.get('eager_preferred_steps') if isinstance(c, dict) else None}")
P
for rep in 1 2; do
  for cfg in metric cfg3 cfg2 cfg1; do
    for g in "" "--graph-always"; do
      echo -n "$cfg $g: " >> $L
      timeout 600 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-reference-lists $g 2>/dev/null | python /tmp/brief2.py >> $L
    done
  done
  for g in "" "--graph-always"; do
    echo -n "views-slice 0/8 $g: " >> $L
    timeout 600 python bench.py --emulate-shard 0/8 --shard views --ar-chunks 4 --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists $g 2>/dev/null | python /tmp/brief2.py >> $L
  done
done
timeout 1500 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "graph or captured or replay or bench or emulated or parts" >> $L 2>&1
cut -c1-200 $L | tail -40
