"""Builds profiles/traffic_<round>.json (default r04) from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, "HBM" and the TCC slot table):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <fetch_dir> --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <write_dir> --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists
    python profiles/make_traffic.py <fetch_dir> <write_dir> [config] [round]
Unit: the counters are in KiB.  gfx950 correction (same guide): FETCH_SIZE reports half the bytes of wide (16 B per
lane) coalesced streaming reads, so it is doubled for the kernels whose reads are of that kind (STREAMING below,
calibrated on blur_loss: 771 MB read, 373 MB reported); the row-gather kernels (48-byte rows picked by index) report
~1x their known row bytes and are taken raw.  WRITE_SIZE is taken as is.  Values are averages per launch.
"""
import collections
import csv
import glob
import json
import os
import sys

STAGE_OF = {
    "composite_bwd_kernel": "composite_bwd", "composite_fwd_kernel": "composite_fwd",
    "preprocess_fwd_kernel": "preprocess", "cull_emit_kernel": "duplicate", "cull_count_kernel": "tile_cull(count)",
    "gather_cnt_kernel": "tile_cull(gather)",
    "duplicate_sorted_kernel": "duplicate", "ranges_search_kernel": "ranges", "contrib_reduce_kernel": "geometry_bwd(contrib_reduce)",
    "geometry_bwd_kernel": "geometry_bwd(kernel)", "sort_scatter_kernel": "sort(scatter)",
    "sort_hist_rows_kernel": "sort(hist)", "blur_loss_kernel": "blur_loss", "blur_loss_all_kernel": "blur_loss",
    "adam_kernel": "adam", "dsort_scatter_kernel": "depth_order(scatter)", "dsort_hist_kernel": "depth_order(hist)",
}
STREAMING = {"blur_loss", "sort(hist)", "sort(scatter)", "adam"}
# round 6: every kernel of the step, and the stage of bench.py's `stages` it belongs to -- `_per_step` sums ALL launches of
# a stage's kernels over the profiled steps and divides by the step count (= launches of the compositing backward, which
# runs once per step under DGS_BWD_OVERLAP=0); bench.py divides by its own per-step stage time: counter-based GB/s
STAGE_OF.update({
    "scan_reduce_kernel": "scan(aux)", "scan_top_kernel": "scan(aux)", "scan_apply_kernel": "scan(aux)",
    "scan_apply_self_kernel": "scan(aux)", "colscan_chunk_kernel": "sort(aux)", "colscan_top_kernel": "sort(aux)",
    "dsort_colscan_chunk_kernel": "depth_order(aux)", "dsort_colscan_top_kernel": "depth_order(aux)",
    "dsort_copy_if_wide_kernel": "depth_order(aux)", "pose_grad_reduce_kernel": "geometry_bwd(pose)",
})
BENCH_STAGE = {
    "preprocess": "preprocess", "scan(aux)": "scan", "duplicate": "duplicate", "sort(hist)": "sort", "sort(scatter)": "sort",
    "sort(aux)": "sort", "ranges": "ranges", "composite_fwd": "composite_fwd", "composite_bwd": "composite_bwd",
    "geometry_bwd(kernel)": "geometry_bwd", "geometry_bwd(pose)": "geometry_bwd",
    "geometry_bwd(contrib_reduce)": "contrib_reduce", "depth_order(hist)": "depth_order",
    "depth_order(scatter)": "depth_order", "depth_order(aux)": "depth_order", "tile_cull(count)": "tile_cull",
    "tile_cull(gather)": "tile_cull", "blur_loss": "(loss image)", "adam": "(adam)",
}


def per_kernel(d, counter, totals=False):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:      # the newest pass only (gpurun merges, it does not replace, earlier passes' files)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            key = next((v for k, v in STAGE_OF.items() if name.startswith(k.split("<")[0]) and
                        ("<" not in k or k in name)), None)
            if key is None:
                continue
            tot[key] += float(r["Counter_Value"]) * 1024.0
            cnt[key] += 1
    if totals:
        return dict(tot), dict(cnt)
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_config": sys.argv[3] if len(sys.argv) > 3 else "metric",
           "_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 "
                      "bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-reference-lists; see profiles/make_traffic.py",
           "_unit": "HBM-side bytes per launch = FETCH_SIZE*1024 (x2 for the 16-B/lane streaming kernels: "
                    + ", ".join(sorted(STREAMING)) + ") + WRITE_SIZE*1024",
           "_detail": {}}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, 0.0) * (2.0 if k in STREAMING else 1.0)
        w = write.get(k, 0.0)
        out["_detail"][k] = {"fetch_bytes": int(f), "write_bytes": int(w)}
        out[k] = int(f + w)
    # per step and bench stage: every launch counted
    ftot, fcnt = per_kernel(sys.argv[1], "FETCH_SIZE", totals=True)
    wtot, wcnt = per_kernel(sys.argv[2], "WRITE_SIZE", totals=True)
    steps_f, steps_w = max(fcnt.get("composite_bwd", 1), 1), max(wcnt.get("composite_bwd", 1), 1)
    # Launches per step: the profiled command also runs forwards outside its steps (two probes, the forward-only region),
    # so a forward-chain kernel is counted per launch of the compositing forward and a backward-chain kernel per launch of
    # the compositing backward (one of each per step); the two kernels outside the rasteriser run once per step.
    FORWARD = {"preprocess", "scan(aux)", "duplicate", "sort(hist)", "sort(scatter)", "sort(aux)", "ranges", "composite_fwd",
               "depth_order(hist)", "depth_order(scatter)", "depth_order(aux)", "tile_cull(count)", "tile_cull(gather)"}
    per_step = collections.defaultdict(float)
    for tot, cnt, stream_x2 in ((ftot, fcnt, True), (wtot, wcnt, False)):
        n_fwd, n_bwd = max(cnt.get("composite_fwd", 1), 1), max(cnt.get("composite_bwd", 1), 1)
        for k, v in tot.items():
            per_launch = v / cnt[k] * (2.0 if (stream_x2 and k in STREAMING) else 1.0)
            launches = cnt[k] / (n_fwd if k in FORWARD else n_bwd)
            per_step[BENCH_STAGE.get(k, "(other)")] += per_launch * max(1, round(launches))
    out["_per_step"] = {k: int(v) for k, v in sorted(per_step.items())}
    out["_per_step_total"] = int(sum(per_step.values()))
    out["_per_step_note"] = ("HBM-side bytes per training step by bench.py stage, all launches of the stage's kernels "
                             f"(profiled steps: {steps_f} in the FETCH pass, {steps_w} in the WRITE pass); '(loss image)' and "
                             "'(adam)' are the step's two kernels outside the rasteriser's stage timers")
    if len(sys.argv) > 5:
        out["_sh_degree"] = int(sys.argv[5])
    rnd = sys.argv[4] if len(sys.argv) > 4 else "r03"
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"traffic_{rnd}.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
