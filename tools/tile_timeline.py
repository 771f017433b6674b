"""How long is the tail of the wave-per-tile schedule of the compositing backward?  

Runs the metric step (eagerly) on the diagnostic build variants/libdgs_timeline.so (composite.hip with -DDGS_TIMELINE=1:
every wave stores its start / end time, s_memrealtime at 100 MHz, by tile) and integrates the number of resident waves
over the launch:

    span         first start -> last end
    busy         sum of the waves' own durations
    peak         largest number of waves resident at once (the occupancy the launch reaches)
    efficiency   busy / (peak * span): 1 = every slot busy from the first cycle to the last
    ramp / tail  time until 90 % of the peak is first reached / time after it is last held
    lost         span - busy / peak: what a perfectly packed schedule of the same waves would save

    DGS_LIB_PATH=variants/libdgs_timeline.so python tools/tile_timeline.py [--config metric] [--json out.json]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="metric")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    if "timeline" not in os.environ.get("DGS_LIB_PATH", ""):   # (any build with -DDGS_TIMELINE=1)
        raise SystemExit("set DGS_LIB_PATH=variants/libdgs_timeline.so (tools/build_variant.sh timeline composite.hip "
                         "deblurgs_amd/csrc/composite.hip -DDGS_TIMELINE=1)")
    os.environ["DGS_BWD_OVERLAP"] = "0"      # one launch over all subframes: the buffer is indexed by the launch's own tiles
    import numpy as np
    import torch
    from deblurgs_amd import _lib, synthetic
    from deblurgs_amd.cloud import GaussianCloud
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    from deblurgs_amd.training import TrainingLoop, default_optimization_params
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    sc = synthetic.make_config(a.config, seed=0)
    P, W, H, K = sc["P"], sc["W"], sc["H"], sc["K"]
    C = synthetic.CONFIGS[a.config]["C"]
    cloud = GaussianCloud.from_scene(sc, dev)
    ref = RefCamera(W, H, sc["FoVx"], sc["FoVy"], device=dev)
    gt = torch.rand((1, 3, H, W), generator=torch.Generator().manual_seed(1234)).to(dev)
    m = CameraMotionModule(ref, gt, curve_order=C, num_subframes=K, device=dev)
    traj = synthetic.make_trajectory(K, C, sc["projection_matrix"], seed=0)
    with torch.no_grad():
        m._trans._control_points.copy_(torch.from_numpy(traj["ctrl_trans"])[None].to(dev))
        m._rot._control_points.copy_(torch.from_numpy(traj["ctrl_rot"])[None].to(dev))
    m.link_gaussian(cloud)
    far = 10 ** 9
    opt = default_optimization_params(iterations=far, curve_start_iter=1, curve_end_iter=far, densify_from_iter=far,
                                      densify_until_iter=far, opacity_reset_interval=far)
    loop = TrainingLoop(cloud, m, opt, cameras_extent=1.0, spatial_lr_scale=1.0, log_losses=False, graph=False)
    for g in cloud.optimizer.param_groups:
        g["lr"] *= 1e-6
    cloud.xyz_scheduler_args = lambda it: 0.00016 * 1e-6
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    buf = torch.zeros(3 * K * T, dtype=torch.int64, device=dev)
    for it in range(1, 5):
        loop.step(it, 0)
    torch.cuda.synchronize()
    assert L.dgs_debug_set_timeline(ctypes.c_void_p(buf.data_ptr())) == 0
    loop.step(5, 0)
    torch.cuda.synchronize()
    L.dgs_debug_set_timeline(None)
    tl = buf.cpu().numpy().reshape(K * T, 3).astype(np.int64)
    ok = tl[:, 1] > 0
    st, en = tl[ok, 0], tl[ok, 1]
    t0 = st.min()
    st, en = (st - t0) * 0.01, (en - t0) * 0.01          # microseconds (100 MHz)
    span = float(en.max())
    busy = float((en - st).sum())
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.lexsort((-ev[:, 1], ev[:, 0]))]
    conc = np.cumsum(ev[:, 1])
    peak = float(conc.max())
    hi = conc >= 0.9 * peak
    ramp = float(ev[np.argmax(hi), 0])
    last_hi = float(ev[len(hi) - 1 - np.argmax(hi[::-1]), 0])
    dur = en - st
    # which XCD ran the tile (XCC_ID register) against the one the block index implies (block b -> XCD b % 8), and the tile
    # run it belongs to (composite.hip: eight contiguous runs, XCD x starts on run x and helps with the others afterwards)
    hw_xcc, by_block, slot = (tl[ok, 2] >> 8) & 0xF, tl[ok, 2] & 0xFF, tl[ok, 2] >> 16
    # per wave slot: the tiles it composited in time order -> the gaps between them (ticket latency the wave sat through)
    order = np.lexsort((st, slot))
    s_sorted, st_s, en_s = slot[order], st[order], en[order]
    same = s_sorted[1:] == s_sorted[:-1]
    gaps = (st_s[1:] - en_s[:-1])[same]
    tiles_per_slot = np.bincount(slot.astype(np.int64))
    tiles_per_slot = tiles_per_slot[tiles_per_slot > 0]
    # durations by when the tile started (tenths of the span)
    decile = np.minimum((st / span * 10).astype(int), 9)
    dur_by_decile = [round(float(dur_all.mean()), 1) if (dur_all := (en - st)[decile == d]).size else None for d in range(10)]
    nblk = (K * T + 3) // 4
    per_run = (nblk + 7) // 8 * 4
    run = np.nonzero(ok)[0] // per_run
    xcd_end = [float(en[hw_xcc == x].max()) for x in range(16) if (hw_xcc == x).any()]
    out = {"config": a.config, "waves": int(ok.sum()), "span_us": round(span, 1), "busy_wave_us": round(busy, 1),
           "peak_resident_waves": int(peak), "mean_resident_waves": round(busy / span, 1),
           "efficiency_busy_over_peak_times_span": round(busy / (peak * span), 4),
           "ramp_to_90pct_us": round(ramp, 1), "tail_after_90pct_us": round(span - last_hi, 1),
           "lost_vs_perfect_packing_us": round(span - busy / peak, 1),
           "lost_fraction_of_span": round((span - busy / peak) / span, 4),
           "wave_duration_us": {"mean": round(float(dur.mean()), 1), "p50": round(float(np.median(dur)), 1),
                                "p99": round(float(np.quantile(dur, 0.99)), 1), "max": round(float(dur.max()), 1)},
           "gap_between_a_waves_tiles_us": ({"mean": round(float(gaps.mean()), 2), "p50": round(float(np.median(gaps)), 2),
                                             "p99": round(float(np.quantile(gaps, 0.99)), 2), "max": round(float(gaps.max()), 2),
                                             "sum_over_busy": round(float(gaps.sum() / busy), 4)} if gaps.size else None),
           "tiles_per_wave_slot": {"min": int(tiles_per_slot.min()), "mean": round(float(tiles_per_slot.mean()), 1),
                                   "max": int(tiles_per_slot.max()), "slots": int(tiles_per_slot.size)},
           "mean_wave_duration_us_by_start_decile": dur_by_decile,
           "xcc_id_equals_block_index_mod_8": float((hw_xcc == by_block).mean()),
           "tiles_composited_by_their_home_xcd": float((by_block == run).mean()),
           "xcd_finish_us": [round(x, 1) for x in xcd_end],
           "xcd_finish_spread_us": round(max(xcd_end) - min(xcd_end), 1)}
    print(json.dumps(out, indent=1))
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
