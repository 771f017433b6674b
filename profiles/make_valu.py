"""Builds profiles/valu_<round>.json: the VALU (and LDS) side of the roofline for the kernels HBM does not bound.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE \
        SQ_LDS_BANK_CONFLICT -d <dir> --output-format csv \
        -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists --no-graph
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -d <dir2> ... -- tools/valu_rate 8
    python profiles/make_valu_peak.py <dir2> profiles/valu_classes_<round>.txt > profiles/valu_peak_<round>.json
    python tools/isa_census.py --json profiles/isa_census_<round>.json
    python profiles/make_valu.py <dir> [config] [round] [R_total]

Per kernel (all measured, no nominal clock): cycles = GRBM_GUI_ACTIVE / 8 XCDs,
  ipc_per_simd          = SQ_INSTS_VALU / (1024 SIMDs x cycles)            wave64 VALU instructions per cycle and SIMD
  frac_of_peak_issue    = ipc_per_simd / 0.5                               the guide's peak: one wave64 fp32 VALU
                          instruction per 2 cycles on a SIMD-32 (confirmed: profiles/valu_peak_<round>.json reaches 0.44-0.47)
  lds_busy_frac         = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles)           LDS-array cycles (a store's address + data
                          transfer, 2 cycles per source dword, is NOT in this counter)
and, for the two compositing kernels, issue_weighted_frac: the dynamic instruction mix (ISA census per pass / per list
entry / per batch x the numbers of entries R, batches ~ R / 64 and passes, the latter solved from the measured
instruction total) weighted with the issue cost of each class = 1 / (instructions per cycle and SIMD that class reaches
in tools/valu_rate UNDER THE SAME COUNTERS, 8 waves per SIMD), divided by the SIMD-cycles of the launch.  The class costs
include the micro-benchmark's own loop overhead, so the weighted fraction is an upper estimate of the time the VALU is
issuing; it is below 1 for every kernel (round 3's s_memtime-based costs were not: its figures above 1 are withdrawn).
"""
import collections, csv, glob, json, os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
rnd = sys.argv[3] if len(sys.argv) > 3 else "r04"
R_total = float(sys.argv[4]) if len(sys.argv) > 4 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
for f in files[-1:]:          # the newest pass only
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))

# ---- issue cost per class: cycles per wave64 instruction on one SIMD at saturation, from the PMC figures of tools/valu_rate
cost = {}
peak_file = os.path.join(HERE, f"valu_peak_{rnd}.json")
if os.path.exists(peak_file):
    peak = json.load(open(peak_file))
    rate = lambda name: peak[name]["pmc_instr_per_cycle_simd"]
    cost = {"valu_plain": 1.0 / rate("v_fma_f32 (3 VGPR sources)"), "valu_trans": 1.0 / rate("v_exp_f32"),
            "valu_dpp": 1.0 / rate("v_add_f32_dpp"), "valu_mov": 1.0 / rate("v_mov_b32"),
            "valu_readlane": 1.0 / rate("v_readlane_b32"), "valu_quarter": 1.0 / rate("v_mad_u64_u32"),
            "valu_pk": 1.0 / rate("v_pk_fma_f32"), "valu_permlane": 1.0 / rate("v_add_f32_dpp")}
    cost["valu_cmp"] = cost["valu_select"] = 1.0 / rate("v_cmp+v_cndmask (2 instr)")
census = {}
cen_file = os.path.join(HERE, f"isa_census_{rnd}.json")
if os.path.exists(cen_file):
    census = json.load(open(cen_file))

out = {"_config": sys.argv[2] if len(sys.argv) > 2 else "metric",
       "_source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE "
                  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline "
                  "--no-reference-lists --no-graph (metric config); class costs: "
                  f"profiles/valu_peak_{rnd}.json (tools/valu_rate under the same counters); mix: profiles/isa_census_{rnd}.json",
       "_unit": "per launch; GRBM_GUI_ACTIVE is summed over the 8 XCDs; ipc_per_simd = SQ_INSTS_VALU / (1024 SIMDs * "
                "GRBM_GUI_ACTIVE / 8); frac_of_peak_issue = ipc_per_simd / 0.5; issue_weighted_frac = sum_class(dynamic count * "
                "PMC-measured issue cost) / (1024 SIMDs * cycles) <= 1; lds_busy_frac = SQ_LDS_IDX_ACTIVE / (256 CUs * cycles)",
       "_class_cost_cycles": {k: round(v, 2) for k, v in cost.items()}}
for k, c in agg.items():
    if not any(x in k for x in ("composite", "geometry_bwd_kernel", "preprocess", "contrib", "cull_count_kernel", "cull_emit_kernel")):
        continue
    v = {n: sum(x) / len(x) for n, x in c.items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    e = {"valu_insts": int(v["SQ_INSTS_VALU"]), "salu_insts": int(v["SQ_INSTS_SALU"]), "lds_insts": int(v["SQ_INSTS_LDS"]),
         "waves": int(v["SQ_WAVES"]), "cycles_per_xcd": int(cyc), "ipc_per_simd": round(v["SQ_INSTS_VALU"] / (1024 * cyc), 3),
         "frac_of_peak_issue": round(v["SQ_INSTS_VALU"] / (1024 * cyc) / 0.5, 3)}
    if "SQ_LDS_IDX_ACTIVE" in v:
        e["lds_array_cycles"] = int(v["SQ_LDS_IDX_ACTIVE"])
        e["lds_bank_conflict_cycles"] = int(v.get("SQ_LDS_BANK_CONFLICT", 0))
        e["lds_busy_frac"] = round(v["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 3)
    # (round 6: the forward has a second template flag, KEEP = state for a backward; the training step runs <.., true>)
    ck = "composite_fwd<false,true>" if k.startswith("composite_fwd_kernel<false, true>") else (
        "composite_fwd<true,true>" if k.startswith("composite_fwd_kernel<true, true>") else (
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
                 issue_weighted_frac=round(weighted / (1024 * cyc), 3),
                 mean_issue_cost_cycles=round(weighted / max(v["SQ_INSTS_VALU"], 1), 2))
    out[k] = e
json.dump(out, open(os.path.join(HERE, f"valu_{rnd}.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
