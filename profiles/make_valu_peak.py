"""Counter-side check of tools/valu_rate: per instruction class, wave64 VALU instructions per core
clock cycle and SIMD from the PMC counters of the micro-benchmark's own kernels, next to the figure the tool derives from
s_memtime.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -d <dir> --output-format csv -- tools/valu_rate 8
    python profiles/make_valu_peak.py <dir> profiles/valu_classes_r04.txt > profiles/valu_peak_r04.json

ipc = SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).  The counter figure includes the loop's own few VALU-free
scalar instructions, the launch ramp and the tail, so it sits a little BELOW the steady-state s_memtime figure."""
import collections, csv, glob, json, os, re, sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"valu_rate_k<\(?(?:Cls\))?(\d+)>", r["Kernel_Name"])
        if m:
            agg[int(m.group(1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
tool = {}
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    for line in open(sys.argv[2]):
        m = re.match(r"(.+?)\s+waves/SIMD (\d+)\s+cycles/instr ([\d.]+)\s+instr/cycle/SIMD ([\d.]+)\s+\(kernel valu_rate_k<(\d+)>\)", line)
        if m and int(m.group(2)) == 8:
            tool[int(m.group(5))] = (m.group(1).strip(), float(m.group(4)))
out = {"_source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -- tools/valu_rate 8 (8 waves per SIMD)",
       "_unit": "wave64 VALU instructions per core clock cycle and SIMD; pmc = SQ_INSTS_VALU / (1024 * GRBM_GUI_ACTIVE / 8), "
                "best dispatch; s_memtime = tools/valu_rate's own figure (mean wave cycles)"}
for c in sorted(agg):
    v = agg[c]
    n = min(len(v["SQ_INSTS_VALU"]), len(v["GRBM_GUI_ACTIVE"]))
    ipc = max(v["SQ_INSTS_VALU"][i] / (1024.0 * v["GRBM_GUI_ACTIVE"][i] / 8.0) for i in range(n)) if n else None
    name, t = tool.get(c, (f"class {c}", None))
    out[name] = {"pmc_instr_per_cycle_simd": None if ipc is None else round(ipc, 3), "s_memtime_instr_per_cycle_simd": t,
                 "valu_insts": int(v["SQ_INSTS_VALU"][0]) if n else None, "waves": int(v["SQ_WAVES"][0]) if v.get("SQ_WAVES") else None}
print(json.dumps(out, indent=1))
