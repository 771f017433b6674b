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


def per_kernel(d, counter):
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
    rnd = sys.argv[4] if len(sys.argv) > 4 else "r03"
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"traffic_{rnd}.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
