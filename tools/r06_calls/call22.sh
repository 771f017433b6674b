#!/bin/bash
# kernel time of 303 single-camera inference calls at the metric scene (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/inf_trace -o trace --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/r06_calls/inference_profile.py metric noprof > /tmp/inf.log 2>&1
tail -2 /tmp/inf.log
f=$(find /tmp/inf_trace -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r06/single_camera_kernel_stats.csv
python3 - <<P
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms over the run; per call (303 calls): %.3f ms" % (tot/1e6, tot/1e6/303))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:16]:
    print("%-70s calls %6s  avg %8.1f us  total %7.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
P
