"""stdin: bench.py's output; prints ms/step and the backward-side stage averages of the JSON line."""
import json
import sys

for ln in sys.stdin:
    if ln.startswith("{"):
        d = json.loads(ln)
        st = d.get("stages", {})
        print(f"{d['ms_per_step']:.3f} ms/step  {d['value']:.1f}/s  " +
              "  ".join(f"{k} {st[k]['avg_ms']:.3f}" for k in ("composite_fwd", "composite_bwd", "contrib_reduce", "geometry_bwd")
                        if k in st))
