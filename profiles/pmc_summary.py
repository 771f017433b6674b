"""Aggregates a rocprofv3 --pmc counter_collection.csv per kernel: python profiles/pmc_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, cs in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0))):
    if not any(x in k for x in ("composite", "geometry", "sort", "contrib", "preprocess", "duplicate", "blur", "pose")):
        continue
    print(k, {c: round(v / max(cnt[(k, c)], 1)) for c, v in cs.items()})
