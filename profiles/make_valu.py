"""Builds profiles/valu_<round>.json (default r03): the VALU side of the roofline for the VALU-bound kernels.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d <dir> --output-format csv \
        -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists --no-graph
    tools/valu_rate > profiles/valu_classes_<round>.txt          # issue cost per instruction class, MEASURED cycles
    python tools/isa_census.py --json profiles/isa_census_<round>.json
    python profiles/make_valu.py <dir> [config] [round] [R_total]

Per kernel: wave-level VALU instructions and core-clock cycles of the launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs),
ipc_per_simd = SQ_INSTS_VALU / (1024 SIMDs x cycles) -- and, for the two compositing kernels, the ISSUE-SLOT-WEIGHTED busy
fraction: the dynamic instruction mix (ISA census per pass / per list entry / per batch x the number of entries R, of
batches ~ R / 64 and of passes, the latter solved from the measured instruction total) weighted with the measured issue
cost of each class (cycles per wave64 instruction at saturation, best of the 4- and 8-waves-per-SIMD rows of
tools/valu_rate), divided by the SIMD-cycles of the launch.  1.0 would mean that the VALU pipes never idle; both the
instruction counts and the cycles are measured, no nominal clock enters.
"""
import collections, csv, glob, json, os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
rnd = sys.argv[3] if len(sys.argv) > 3 else "r03"
R_total = float(sys.argv[4]) if len(sys.argv) > 4 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
for f in files[-1:]:          # the newest pass only
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))

# ---- issue cost per class (cycles per instruction on one SIMD at saturation)
cost = {}
cls_file = os.path.join(HERE, f"valu_classes_{rnd}.txt")
if os.path.exists(cls_file):
    best = collections.defaultdict(lambda: 1e9)
    for line in open(cls_file):
        m = re.match(r"(.+?)\s+waves/SIMD (\d+)\s+cycles/instr ([\d.]+)", line)
        if m and int(m.group(2)) in (4, 8):
            best[m.group(1).strip()] = min(best[m.group(1).strip()], float(m.group(3)))
    name_of = {"valu_plain": "v_fma_f32", "valu_trans": "v_exp_f32", "valu_dpp": "v_add_f32_dpp",
               "valu_permlane": "v_permlane32_swap", "valu_mov": "v_mov_b32"}
    cost = {c: best[n] for c, n in name_of.items() if n in best}
    if "v_cmp+v_cndmask (2 instr)" in best:
        cost["valu_cmp"] = cost["valu_select"] = best["v_cmp+v_cndmask (2 instr)"]
census = {}
cen_file = os.path.join(HERE, f"isa_census_{rnd}.json")
if os.path.exists(cen_file):
    census = json.load(open(cen_file))

out = {"_config": sys.argv[2] if len(sys.argv) > 2 else "metric",
       "_source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -- python3 "
                  "bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists --no-graph (metric config); class costs: "
                  f"profiles/valu_classes_{rnd}.txt (tools/valu_rate, measured cycles); mix: profiles/isa_census_{rnd}.json",
       "_unit": "per launch; GRBM_GUI_ACTIVE is summed over the 8 XCDs; ipc_per_simd = SQ_INSTS_VALU / (1024 SIMDs * "
                "GRBM_GUI_ACTIVE / 8); weighted_busy_frac = sum_class(dynamic count * measured issue cost) / (1024 SIMDs * cycles)",
       "_class_cost_cycles": {k: round(v, 2) for k, v in cost.items()}}
for k, c in agg.items():
    if not any(x in k for x in ("composite", "geometry_bwd_kernel", "preprocess", "contrib", "cull_count_kernel", "cull_emit_kernel")):
        continue
    v = {n: sum(x) / len(x) for n, x in c.items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    e = {"valu_insts": int(v["SQ_INSTS_VALU"]), "salu_insts": int(v["SQ_INSTS_SALU"]), "lds_insts": int(v["SQ_INSTS_LDS"]),
         "waves": int(v["SQ_WAVES"]), "cycles_per_xcd": int(cyc), "ipc_per_simd": round(v["SQ_INSTS_VALU"] / (1024 * cyc), 3)}
    ck = "composite_fwd<false>" if k.startswith("composite_fwd_kernel<false>") else (
        "composite_fwd<true>" if k.startswith("composite_fwd_kernel<true>") else (
            "composite_bwd<true,false>" if k.startswith("composite_bwd_kernel<true, false>") else None))
    if ck in census and cost and R_total:
        cen = census[ck]
        valu = lambda d: sum(n for cl, n in d.items() if cl.startswith("valu"))
        wc = lambda d: sum(n * cost.get(cl, cost["valu_plain"]) for cl, n in d.items() if cl.startswith("valu"))
        entries, batches = R_total, R_total / 64.0 + 0.5 * e["waves"]
        fixed = entries * valu(cen["per_entry_outside_passes"]) + batches * valu(cen["per_batch_outside_entry_loop"])
        # the census counts every static instruction of the entry loop; v_mov zero-fills and branches not taken make the
        # dynamic count smaller: scale the fixed part down if it alone exceeds the measured total
        passes = max((v["SQ_INSTS_VALU"] - fixed) / max(valu(cen["per_pass"]), 1e-9), 0.0)
        weighted = (passes * wc(cen["per_pass"]) + entries * wc(cen["per_entry_outside_passes"]) +
                    batches * wc(cen["per_batch_outside_entry_loop"]))
        e.update(entries=int(entries), passes_estimated=int(passes), passes_per_entry=round(passes / entries, 2),
                 weighted_busy_frac=round(weighted / (1024 * cyc), 3),
                 mean_issue_cost_cycles=round(weighted / max(v["SQ_INSTS_VALU"], 1), 2))
    out[k] = e
json.dump(out, open(os.path.join(HERE, f"valu_{rnd}.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
