"""Predicted 2 / 4 / 8-GPU scaling of the metric step from ONE GPU: the measured part is what a rank
computes between its exchanges (bench.py --emulate-shard r/G: the sharded step of rank r in a one-rank process group on
the real backend, collectives degenerate); the modelled part is the exchanges, priced with SURVEY 8e's link model.

    python tools/predict_scaling.py [--config metric] [--steps 60] [--out gpurun_out/predicted_scaling.json]

Model (every assumption is in the output):
  * xGMI: G GPUs fully connected, one link per pair, `link` GB/s per direction (default 0.8 x 76.8: 153.6 GB/s per link
    is the bidirectional figure of MI355X_MICROARCH.md; the 0.8 is an assumption -- no xGMI transfer has been timed);
  * all-reduce of B bytes, "direct" (reduce-scatter + all-gather over point-to-point transfers, sharding.p2p_allreduce_multi_
    or an RCCL that uses every link): 2 * (B / G) / link; "ring" (one ring, one link busy per hop): 2 (G-1)/G * B / link.
    RCCL on a fully connected node builds several rings over different links and should land between the two;
  * "views": the bucket's all-reduce is issued in `chunks` Gaussian-index chunks behind the per-Gaussian half of the
    backward; exposed time = max(t_ar / chunks, t_ar - t_geometry * (chunks - 1) / chunks);
  * "subframes": before the backward, the partial blur image [3,H,W] is all-reduced and one boundary subframe travels to
    each neighbour (blur_bytes / link, both directions at once); after it the whole bucket's all-reduce is exposed (the
    per-Gaussian kernel of 1-2 subframes is too short to hide anything).
  * "mesh" (round 6, G >= 4): G/2 rows of two ranks, a row splits ONE view's subframes (the measured slice of a 2-rank
    "subframes" step), exchanges its blur image and boundary frame inside the pair, and the bucket is all-reduced over all
    G ranks, fully exposed (as for "subframes"); the job renders G/2 views per step.
  THE LINK MODEL IS UNVERIFIED: no xGMI transfer has ever been timed by this project (one-GPU boxes only); every predicted
  speed-up below inherits that.
  speed-up = renders per second of G GPUs / renders per second of the single-GPU step (bench.py's default: TrainingLoop's own graph policy) measured in the
  same invocation on the same box.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_bench(extra, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DGS_DIST_BACKEND", "DGS_DIST_ONE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}")
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="metric")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--link-GBps", type=float, default=0.8 * 76.8, help="achievable GB/s per xGMI link and direction")
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "predicted_scaling.json"))
    a = ap.parse_args()
    from deblurgs_amd.sharding import shard_range
    common = ["--config", a.config, "--steps", str(a.steps), "--warmup", str(a.warmup)]
    base = run_bench(common + ["--no-cpu-baseline", "--no-reference-lists"])
    K = int(base["config"]["workload"].split("K=")[1].split(" ")[0])
    t1 = base["ms_per_step"]
    doc = {"config": a.config, "K": K, "single_gpu": {"ms_per_step": t1, "value": base["value"],
                                                       "graph": base["config"].get("graph")},
           "model": {"link_GBps_per_direction": a.link_GBps, "chunks": a.chunks, "link_model": "UNVERIFIED",
                     "assumptions": "see the docstring of tools/predict_scaling.py; the xGMI link model is UNVERIFIED: no "
                                    "xGMI transfer has been timed"},
           "rows": []}
    views = run_bench(common + ["--shard", "views", "--ar-chunks", str(a.chunks), "--emulate-shard", "0/8"])
    ev = views["emulated_shard"]
    # what the chunked all-reduce can hide behind: the per-Gaussian kernel's chunks of one step (the pose sums follow them)
    t_geom = views["stages"].get("geometry_bwd", {}).get("ms_per_step", 0.0)
    B, blur = ev["bucket_bytes"], ev["blur_bytes"]
    doc["views_slice"] = {"ms_per_step": ev["ms_per_step"], "eager_ms_per_step": ev.get("eager_ms_per_step"),
                          "geometry_bwd_ms": t_geom, "graph": ev.get("graph")}
    link = a.link_GBps * 1e9

    def ar(bytes_, G):
        return {"direct": 2.0 * (bytes_ / G) / link * 1e3, "ring": 2.0 * (G - 1) / G * bytes_ / link * 1e3}
    for G in [int(x) for x in a.worlds.split(",")]:
        # ---- views (weak scaling): every rank a whole view
        t_ar = ar(B, G)
        row_v = {"G": G, "sharding": "views", "slice_ms": ev["ms_per_step"], "allreduce_ms": t_ar}
        for kind in ("direct", "ring"):
            exposed = max(t_ar[kind] / a.chunks, t_ar[kind] - t_geom * (a.chunks - 1) / a.chunks)
            step = ev["ms_per_step"] + exposed
            row_v[kind] = {"exposed_comm_ms": round(exposed, 3), "step_ms": round(step, 3),
                           "renders_per_s": round(G * K / step * 1e3, 1), "speedup": round(G * t1 / step, 2)}
        doc["rows"].append(row_v)
        # ---- subframes (strong scaling): the rank with the most subframes sets the pace
        sizes = [shard_range(K, r, G)[1] - shard_range(K, r, G)[0] for r in range(G)]
        r_slow = max(range(G), key=lambda r: (sizes[r], -r))
        sub = run_bench(common + ["--shard", "subframes", "--ar-chunks", str(a.chunks), "--emulate-shard",
                                  f"{r_slow}/{G}"])["emulated_shard"]
        t_blur = ar(blur, G)
        row_s = {"G": G, "sharding": "subframes", "slowest_rank": r_slow, "subframes_of_that_rank": sizes[r_slow],
                 "slice_ms": sub["ms_per_step"], "slice_eager_ms": sub.get("eager_ms_per_step"), "allreduce_ms": t_ar,
                 "blur_allreduce_ms": t_blur, "boundary_ms": blur / link * 1e3}
        for kind in ("direct", "ring"):
            comm = t_blur[kind] + blur / link * 1e3 + t_ar[kind]
            step = sub["ms_per_step"] + comm
            row_s[kind] = {"exposed_comm_ms": round(comm, 3), "step_ms": round(step, 3),
                           "renders_per_s": round(K / step * 1e3, 1), "speedup": round(t1 / step, 2)}
        doc["rows"].append(row_s)
        # ---- mesh: G/2 views per step, each view's subframes split over a pair of ranks
        if G >= 4 and G % 2 == 0:
            sizes2 = [shard_range(K, r, 2)[1] - shard_range(K, r, 2)[0] for r in range(2)]
            r2 = max(range(2), key=lambda r: (sizes2[r], -r))
            if "pair_slice" not in doc:
                doc["pair_slice"] = run_bench(common + ["--shard", "subframes", "--ar-chunks", str(a.chunks), "--emulate-shard",
                                                        f"{r2}/2"])["emulated_shard"]
            pair = doc["pair_slice"]
            t_blur2 = ar(blur, 2)
            row_m = {"G": G, "sharding": f"mesh {G // 2}x2", "views_per_step": G // 2, "slice_ms": pair["ms_per_step"],
                     "allreduce_ms": t_ar, "blur_allreduce_ms": t_blur2, "boundary_ms": blur / link * 1e3}
            for kind in ("direct", "ring"):
                comm = t_blur2[kind] + blur / link * 1e3 + t_ar[kind]
                step = pair["ms_per_step"] + comm
                row_m[kind] = {"exposed_comm_ms": round(comm, 3), "step_ms": round(step, 3),
                               "renders_per_s": round((G // 2) * K / step * 1e3, 1),
                               "speedup": round((G // 2) * t1 / step, 2)}
            doc["rows"].append(row_m)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(doc, open(a.out, "w"), indent=1)
    print(f"single GPU: {t1:.3f} ms per K={K} step ({base['value']:.1f} renders/s)   [xGMI link model UNVERIFIED: "
          f"{a.link_GBps:.1f} GB/s per link and direction assumed]")
    print("| G | sharding | rank's own work (ms) | exchanges direct / ring (ms) | step direct / ring (ms) | speed-up direct / ring |")
    print("|---|---|---|---|---|---|")
    for r in doc["rows"]:
        print(f"| {r['G']} | {r['sharding']} | {r['slice_ms']:.2f} | {r['direct']['exposed_comm_ms']:.2f} / "
              f"{r['ring']['exposed_comm_ms']:.2f} | {r['direct']['step_ms']:.2f} / {r['ring']['step_ms']:.2f} | "
              f"{r['direct']['speedup']:.2f} / {r['ring']['speedup']:.2f} |")


if __name__ == "__main__":
    main()
