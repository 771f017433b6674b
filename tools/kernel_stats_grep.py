"""Prints the rows of a rocprofv3 --stats kernel_stats CSV whose kernel name starts with one of the given prefixes:
python tools/kernel_stats_grep.py <dir> <steps> prefix [prefix ...]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2])
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if any(n.startswith(p) for p in sys.argv[3:]):
        print(f"{n:34s} calls {r['Calls']:>5s}  avg_us {float(r['AverageNs']) / 1e3:8.1f}  ms/step {float(r['TotalDurationNs']) / 1e6 / steps:7.4f}")
