#!/bin/bash
# round 6 call 19: the forward-only region with the reference's own call shape (render() on one camera) next to the K-fused one
mkdir -p gpurun_out/r06
for cfg in metric cfg2; do
  timeout 600 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-reference-lists 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); f=d['forward_only']
print('$cfg', d['ms_per_step'], 'fused:', f['value'], f['ms_per_call'], 'single:', f['single_camera']['value'], f['single_camera']['ms_per_call'])"
done | tee gpurun_out/r06/single_camera.log
