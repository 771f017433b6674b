"""Builds profiles/valu_r02.json from one rocprofv3 PMC pass:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d <dir> --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists
    python profiles/make_valu.py <dir> [config]
"""
import collections, csv, glob, json, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
for f in files[-1:]:          # the newest pass only
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_config": sys.argv[2] if len(sys.argv) > 2 else "metric",
       "_source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -- python3 "
                  "bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists (metric config)",
       "_unit": "per launch; GRBM_GUI_ACTIVE is summed over the 8 XCDs; ipc_per_simd = SQ_INSTS_VALU / (1024 SIMDs * "
                "GRBM_GUI_ACTIVE / 8); tools/valu_rate.hip measures 0.32 (4 waves/SIMD) to 0.35 (8 waves/SIMD) wave-instructions "
                "per cycle and SIMD for a pure scalar fp32 FMA stream on this chip (0.5 nominal), 0.21 for packed FMAs"}
for k, c in agg.items():
    if not any(x in k for x in ("composite", "geometry_bwd_kernel", "preprocess", "contrib", "tight_kernel")):
        continue
    v = {n: sum(x) / len(x) for n, x in c.items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    out[k] = {"peak_measured_ipc_per_simd": 0.35, "peak_nominal_ipc_per_simd": 0.5,
              "valu_insts": int(v["SQ_INSTS_VALU"]), "salu_insts": int(v["SQ_INSTS_SALU"]), "lds_insts": int(v["SQ_INSTS_LDS"]),
              "waves": int(v["SQ_WAVES"]), "cycles_per_xcd": int(cyc), "ipc_per_simd": round(v["SQ_INSTS_VALU"] / (1024 * cyc), 3)}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_r02.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
