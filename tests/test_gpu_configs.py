"""BASELINE.json's configurations through the product path EXACTLY AS bench.py RUNS IT -- GaussianCloud raw parameters ->
render_subframes -> rasterize_cloud_subframes (DgsProblem.raw_params = 1), tile culling on, all K subframes in one
fused launch chain -- at full size:

  metric  1M Gaussians, 1920x1080, K=15          (the headline number)
  cfg3    1M Gaussians, 1600x1200, K=15
  cfg5    5M Gaussians, 3840x2160, K=31          (stress: 17+ tile bits, hundreds of millions of duplicates)

For each: size-independent properties of the fused binning at full size (sortedness, range/key consistency, a checksum
of the duplicate lists, key width per rasterizer_impl.cu:306-314), tile_cull = 0 (the reference's lists) against
tile_cull = 1 (the benchmarked lists) bit for bit, and forward + backward against the OpenMP oracle (deterministic,
double accumulation) with the conditioning-aware bars of helpers.assert_grads_close.  The oracle is fed the activated
values the kernels themselves use (dgs_cloud_activations), so radii / tile counts / point lists compare bit for bit."""
import os
import time

import numpy as np
import pytest

from helpers import (CLOUD_KEYS, GRAD_TOL, OracleRun, assert_grads_close, cloud_grads_from_activated,
                     hip_cloud_forward_backward, hip_state_on_device, synthetic)

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4
KEY_BITS_K1 = {"cfg2": 44, "cfg3": 45, "metric": 45, "cfg5": 47}     # 32 + getHigherMsb(T)


def kernel_activated_scene(sc):
    """The scene with scales / rotations / opacities replaced by what the raw-parameter kernels compute from the
    cloud's raw tensors (log-scale -> exp, normalise, clamp): the values the oracle must see for bit-exact integers."""
    from deblurgs_amd.cloud import GaussianCloud
    cloud = GaussianCloud.from_scene(sc, "cuda")
    s, r, o = cloud.device_activations()
    out = dict(sc)
    out["scales"], out["rotations"], out["opacities"] = s.cpu().numpy(), r.cpu().numpy(), o.cpu().numpy()
    return out


def check_fused_binning_properties(st, cull, name):
    """Full-size invariants of the fused K-subframe binning, evaluated on the device."""
    import torch
    K, T, R = st["K"], st["T"], st["R"]
    keys, pl = st["keys"], st["point_list"].long()
    assert R == keys.shape[0] and R > 0
    tile = keys >> 32
    assert bool((tile >= 0).all()) and int(tile.max()) < K * T
    # stable sort on the tile bits of a list generated in (k, depth, index) order: the tile word is non-decreasing, and
    # inside a tile the low word (depth bits with the reference's lists, emission index with tile culling) ascends too
    assert bool((keys[1:] >= keys[:-1]).all()), "sorted keys"
    counts = torch.bincount(tile, minlength=K * T)
    rng = st["ranges"].long()
    assert torch.equal(rng[:, 1] - rng[:, 0], counts), "tile ranges vs keys"
    nonempty = counts > 0
    starts = torch.cumsum(counts, 0) - counts
    assert torch.equal(rng[nonempty, 0], starts[nonempty]), "range starts"
    # every duplicate belongs to a visible (k, Gaussian) whose rectangle contains the tile
    k_of = tile // T
    rad = st["radii"].long()[k_of, pl]
    assert bool((rad > 0).all())
    rows = st["rows"][k_of, pl]
    tl = tile - k_of * T
    gx = (st["W"] + 15) // 16
    tx, ty = (tl % gx).float(), (tl // gx).float()
    x, y, r = rows[:, 0], rows[:, 1], rad.float()
    inside = (tx * 16 <= x + r + 15) & (tx * 16 + 15 >= x - r - 15) & (ty * 16 <= y + r + 15) & (ty * 16 + 15 >= y - r - 15)
    assert bool(inside.all()), "a duplicate outside its Gaussian's tile rectangle"
    # checksums: per (k, Gaussian) duplicate counts against the counts the expansion was sized with
    per_pair = torch.bincount(k_of * st["P"] + pl, minlength=K * st["P"])
    if cull:
        u = keys & 0xFFFFFFFF
        assert int(u.max()) == R - 1 and int(torch.bincount(u, minlength=R).max()) == 1, "emission indices = permutation"
        assert bool((per_pair <= st["tiles_touched"].long().reshape(-1)).all())
    else:
        assert torch.equal(per_pair, st["tiles_touched"].long().reshape(-1)), "duplicates per (k, Gaussian)"
        assert R == int(st["tiles_touched"].long().sum())
    from oracle import oracle
    assert st["sort_bits"] == 32 + oracle.higher_msb(K * T), name


DISAGREE = {}      # per call: pixels whose colour / n_contrib differ from the oracle's, masked or not (the parity A/B's figure)


def forward_against_oracle(st_color, st_depth, st_ncontrib, st_finalT, run, sc, ks):
    worst_frac = 0.0
    DISAGREE.clear()
    DISAGREE.update(pixels=0, colour_off=0, colour_off_unmasked=0, n_contrib_off=0, masked=0)
    for k in ks:
        o, un = run.states[k], run.unstable[k]
        frac = float(un.mean())
        worst_frac = max(worst_frac, frac)
        off = np.abs(st_color[k] - o["color"]).max(axis=0) > IMG_TOL
        DISAGREE["pixels"] += int(un.size)
        DISAGREE["masked"] += int(un.sum())
        DISAGREE["colour_off"] += int(off.sum())
        DISAGREE["colour_off_unmasked"] += int((off & ~un).sum())
        if st_ncontrib is not None:
            DISAGREE["n_contrib_off"] += int((st_ncontrib[k] != o["n_contrib"]).sum())
        assert frac < 1e-3, f"exempt pixel fraction {frac}"          # exempted pixels are counted, not assumed rare
        dc = np.abs(st_color[k] - o["color"]).max(axis=0)
        dd = np.abs(st_depth[k][0] - o["depth"][0]) / sc["z_far"]
        assert dc[~un].max() <= IMG_TOL, f"colour k={k}: {dc[~un].max()}"
        assert dd[~un].max() <= IMG_TOL, f"depth k={k}: {dd[~un].max()}"
        assert dc.max() <= 2e-2 and dd.max() <= 2e-2
        s = ~un.reshape(-1)
        if st_ncontrib is not None:      # positions in the reference's (tile_cull = 0) lists
            assert np.array_equal(st_ncontrib[k][s], o["n_contrib"][s]), "n_contrib"
        assert np.abs(st_finalT[k][s] - o["final_T"][s]).max() <= 1e-5
    return worst_frac


@pytest.mark.parametrize("cfg", ["cfg2", "cfg2_sh3", "metric", "cfg3"])
def test_config_as_benchmarked(gpu, cfg):
    """cfg2_sh3: SURVEY 8's secondary variant (D = 3, M = 16, the upstream-3DGS default; SH paths forward.cu:20-82,
    backward.cu:20-140) at cfg2's size -- preprocess_fwd_kernel<3> and geometry_bwd_kernel<16> as profiles/bench_r05_metric_sh3.json
    times them."""
    import torch
    t0 = time.time()
    marks = []

    def mark(what):       # where the test's time goes (printed at the end; pytest -s)
        torch.cuda.synchronize()
        marks.append((what, time.time()))
    sh3 = cfg.endswith("_sh3")
    cfg = cfg.split("_")[0]
    sc = synthetic.make_config(cfg, sh_degree=3) if sh3 else synthetic.make_config(cfg)
    assert sc["sh"].shape[1] == (16 if sh3 else 9) and sc["sh_degree"] == (3 if sh3 else 2)
    K, P, W, H = sc["K"], sc["P"], sc["W"], sc["H"]
    assert K == (9 if cfg == "cfg2" else 15)
    act = kernel_activated_scene(sc)
    assert np.abs(act["scales"] / sc["scales"] - 1).max() < 5e-6       # exp(log(s)): a few ulps of log(s)
    mark("scene")

    # ---- (1) binning properties at full size, both duplicate rules, and their images bit for bit
    st1 = hip_state_on_device(sc, K, cull=True, raw=True)
    st1.update(P=P, W=W, H=H)
    check_fused_binning_properties(st1, True, cfg)
    st0 = hip_state_on_device(sc, K, cull=False, raw=True, checksum=True)
    st0.update(P=P, W=W, H=H)
    check_fused_binning_properties(st0, False, cfg)
    for key in ("color", "depth", "radii", "final_T", "tiles_touched"):
        assert torch.equal(st0[key], st1[key]), f"tile_cull 0 vs 1: {key}"
    # (n_contrib is a position in the tile's list: with culling it counts the surviving entries only)
    assert bool((st1["n_contrib"] <= st0["n_contrib"]).all())
    assert 0.3 < st1["R"] / st0["R"] < 0.9
    # key width (rasterizer_impl.cu:306-314): one subframe needs 32 + getHigherMsb(T) bits, the fused launch K*T tiles
    from deblurgs_amd import _lib
    assert _lib.layout(P, W, H, 1, 0).sort_bits == KEY_BITS_K1[cfg]
    # the reference's lists are subframe-major: the K=1 reference keys are the fused keys minus k*T in the tile word
    R0 = st0["R"]
    mark("hip states + binning properties")

    # ---- (2) forward of ALL K subframes against the oracle (kernel-activated parameters)
    run = OracleRun(act, K, margin_masks=False)
    # the exempt pixels: where this forward and the oracle's took a different per-pair decision (contributor checksums /
    # last contributors differ) -- NOT every pixel within a margin of a threshold (0.5 % of them at the metric size)
    exempt = run.use_exact_masks(st0["contrib_checksum"].cpu().numpy(), st0["n_contrib"].cpu().numpy())
    assert sum(exempt) < 1e-4 * K * H * W, f"exempt pixels {sum(exempt)} of {K * H * W}"
    mark("oracle forward + exact masks")
    radii = st1["radii"].cpu().numpy()
    tt = st0["tiles_touched"].cpu().numpy().view(np.uint32)
    off = 0
    for k in range(K):
        o = run.states[k]
        assert np.array_equal(radii[k], o["radii"]), f"radii k={k}"
        assert np.array_equal(tt[k], o["tiles_touched"]), f"tiles_touched k={k}"
        Rk = o["num_rendered"]
        assert torch.equal(st0["point_list"][off:off + Rk].cpu(), torch.from_numpy(o["point_list"].view(np.int32)))
        keys_k = st0["keys"][off:off + Rk].cpu().numpy().view(np.uint64) - (np.uint64(k * st0["T"]) << np.uint64(32))
        assert np.array_equal(keys_k, o["keys"]), f"sort keys k={k}"
        off += Rk
    assert off == R0
    frac = forward_against_oracle(st1["color"].cpu().numpy(), st1["depth"].cpu().numpy(),
                                  st0["n_contrib"].cpu().numpy().view(np.uint32), st1["final_T"].cpu().numpy(), run, sc,
                                  range(K))
    fwd_stats = dict(DISAGREE)
    del st0, st1
    torch.cuda.empty_cache()
    mark("forward comparisons")

    # ---- (3) backward as benchmarked (raw parameters, tile culling, K fused) against the oracle.  The metric configuration
    # (and cfg2) with an upstream gradient on ALL K subframes; cfg3 -- same cloud size, another aspect ratio -- on five of
    # its fifteen (first, last, three in between): the per-Gaussian sums are then those subframes', the others' per-subframe
    # outputs must come back as exact zeros, and the suite stays inside the driver's time limit
    bwd_ks = list(range(K)) if cfg != "cfg3" else [0, 3, 7, 11, 14]
    rng = np.random.default_rng(3)
    gC = rng.normal(size=(K, 3, H, W)).astype(np.float32)
    gC, _ = run.mask(gC)
    for k in range(K):
        if k not in bwd_ks:
            gC[k] = 0.0
    hip = hip_cloud_forward_backward(sc, K, gC, cull=True, conic_ks=bwd_ks if len(bwd_ks) < K else None)
    mark("hip backward")
    ora = cloud_grads_from_activated(act, (run if len(bwd_ks) == K else run.subset(bwd_ks)).backward(gC[bwd_ks]))
    mark("oracle backward x3 modes")
    hip_full = hip
    if len(bwd_ks) < K:
        hip = dict(hip)
        for key in ("dL_dmeans2D", "dL_dviewmatrix", "dL_dprojmatrix"):
            rest = np.delete(hip[key], bwd_ks, axis=0)
            assert not rest.any(), f"{key}: subframes without upstream gradient must get exact zeros"
            hip[key] = hip[key][bwd_ks]
    report = []
    assert_grads_close(hip, ora, CLOUD_KEYS, report=report)
    mark("gradient comparisons")
    hip = hip_full
    # ---- (4) the reference's lists give the same gradients bit for bit at this size
    hip0 = hip_cloud_forward_backward(sc, K, gC, cull=False, conic_ks=bwd_ks if len(bwd_ks) < K else None)
    for key in CLOUD_KEYS + ["color", "depth"]:
        assert np.array_equal(hip0[key], hip[key]), f"tile_cull 0 vs 1: {key}"
    print(f"\n[{cfg}{'_sh3' if sh3 else ''}] lib {os.path.basename(os.environ.get('DGS_LIB_PATH', 'libdgs_hip.so'))}: of "
          f"{fwd_stats['pixels']} pixels {fwd_stats['masked']} are exempt (a per-pair decision differs from the oracle's), "
          f"{fwd_stats['colour_off']} differ from the oracle by more than {IMG_TOL} in colour ({fwd_stats['colour_off_unmasked']} "
          f"of them unmasked), {fwd_stats['n_contrib_off']} in n_contrib")
    print(f"[{cfg}] unstable fraction {frac:.2e}; errors (key, hip, oracle-fp32-noise):")
    for r in report:
        print("   ", r)
    mark("reference-lists backward")
    print(f"[{cfg}] {time.time() - t0:.0f} s: " + ", ".join(f"{w} {t - (marks[i - 1][1] if i else t0):.1f}"
                                                           for i, (w, t) in enumerate(marks)))


def test_cfg5_stress_as_benchmarked(gpu):
    """5M Gaussians, 3840x2160, K=31, curve order 5: properties of the benchmarked variant at full size, then eight of the
    subframes forward and the middle subframe's backward against the oracle."""
    import torch
    t0 = time.time()
    sc = synthetic.make_config("cfg5")
    K, P, W, H = sc["K"], sc["P"], sc["W"], sc["H"]
    assert (K, P, W, H) == (31, 5_000_000, 3840, 2160)
    act = kernel_activated_scene(sc)
    st = hip_state_on_device(sc, K, cull=True, raw=True)
    st.update(P=P, W=W, H=H)
    check_fused_binning_properties(st, True, "cfg5")
    from deblurgs_amd import _lib
    assert _lib.layout(P, W, H, 1, 0).sort_bits == KEY_BITS_K1["cfg5"]
    assert st["sort_bits"] == 32 + 20 and st["sort_passes"] == 3        # K*T = 1 004 400 tiles -> 20 tile bits
    # forward: eight of the 31 subframes against the oracle (round 5: three); exempt pixels = where the per-pair decisions
    # differ (contributor checksums of a tile_cull = 0 forward of those subframes)
    ks = [0, 4, 9, 13, K // 2, 19, 25, K - 1]
    mid = ks.index(K // 2)
    sub, sub_raw = dict(act), dict(sc)
    for name in ("viewmatrix", "projmatrix", "campos"):
        sub[name] = act[name][ks]
        sub_raw[name] = sc[name][ks]
    sub_raw["K"] = len(ks)
    run = OracleRun(sub, len(ks), margin_masks=False)
    radii = st["radii"][ks].cpu().numpy()
    for i in range(len(ks)):
        assert np.array_equal(radii[i], run.states[i]["radii"])
    color, depth, final_T = (st[n][ks].cpu().numpy() for n in ("color", "depth", "final_T"))
    R = st["R"]
    del st
    torch.cuda.empty_cache()
    stc = hip_state_on_device(sub_raw, len(ks), cull=False, raw=True, checksum=True)
    assert np.array_equal(stc["color"].cpu().numpy(), color), "tile_cull 0 vs 1 at cfg5"
    exempt = run.use_exact_masks(stc["contrib_checksum"].cpu().numpy(), stc["n_contrib"].cpu().numpy())
    assert sum(exempt) < 1e-4 * len(ks) * H * W, exempt
    frac = forward_against_oracle(color, depth, stc["n_contrib"].cpu().numpy().view(np.uint32), final_T, run, sc,
                                  range(len(ks)))
    del stc
    torch.cuda.empty_cache()
    # backward: upstream gradient on the middle subframe only, so the per-Gaussian sums are that subframe's
    rng = np.random.default_rng(4)
    g_mid = rng.normal(size=(1, 3, H, W)).astype(np.float32)
    g_mid[0][:, run.unstable[mid]] = 0.0
    gC = np.zeros((K, 3, H, W), np.float32)
    gC[K // 2] = g_mid[0]
    hip = hip_cloud_forward_backward(sc, K, gC, cull=True, conic_ks=[K // 2])
    ora = cloud_grads_from_activated(act, run.subset([mid]).backward(g_mid))
    if os.environ.get("DGS_PARITY_ROW"):      # debugging aid: the pixels of one Gaussian that sit closest to the alpha threshold
        gdbg = int(os.environ["DGS_PARITY_ROW"])
        stm = run.states[mid]
        mx, my = (float(t) for t in stm["means2D"][gdbg])
        ca, cb, cc, op = (float(t) for t in stm["conic_opacity"][gdbg])
        rad = int(stm["radii"][gdbg])
        xs = np.arange(max(int(mx) - rad - 1, 0), min(int(mx) + rad + 2, W), dtype=np.float32)
        ys = np.arange(max(int(my) - rad - 1, 0), min(int(my) + rad + 2, H), dtype=np.float32)
        dx = np.float32(mx) - xs[None, :]
        dy = np.float32(my) - ys[:, None]
        power = (np.float32(-0.5) * (np.float32(ca) * dx * dx + np.float32(cc) * dy * dy) - np.float32(cb) * dx * dy).astype(np.float32)
        alpha = np.minimum(np.float32(0.99), np.float32(op) * np.exp(power)).astype(np.float32)
        rel = np.abs(alpha - 1.0 / 255.0) / (1.0 / 255.0)
        order = np.argsort(rel.reshape(-1))[:8]
        print(f"   [row {gdbg}] mean ({mx:.3f}, {my:.3f}) conic ({ca:.5g}, {cb:.5g}, {cc:.5g}) opacity {op:.5g} radius {rad}; "
              f"{int((alpha >= 1 / 255).sum())} pixels reach 1/255")
        for o in order:
            iy, ix = divmod(int(o), xs.shape[0])
            py_, px_ = int(ys[iy]), int(xs[ix])
            terms = 0.5 * abs(ca * dx[0, ix] ** 2) + 0.5 * abs(cc * dy[iy, 0] ** 2) + abs(cb * dx[0, ix] * dy[iy, 0])
            print(f"       pixel ({px_}, {py_}): alpha {alpha[iy, ix]:.9g} (1/255 = {1 / 255:.9g}, rel {rel[iy, ix]:.2e}) power "
                  f"{power[iy, ix]:.6g} largest term {terms:.4g} unstable {bool(run.unstable[mid][py_, px_])} "
                  f"upstream |g| {float(np.abs(g_mid[0][:, py_, px_]).max()):.3g}")
    for key in ("dL_dmeans2D", "dL_dviewmatrix", "dL_dprojmatrix"):
        rest = np.delete(hip[key], K // 2, axis=0)
        assert not rest.any(), f"{key}: subframes without upstream gradient must get exact zeros"
        hip[key] = hip[key][K // 2:K // 2 + 1]
    report = []
    assert_grads_close(hip, ora, CLOUD_KEYS, report=report)
    print(f"\n[cfg5] R = {R}, unstable fraction {frac:.2e}; errors:")
    for r in report:
        print("   ", r)
    print(f"[cfg5] {time.time() - t0:.0f} s")


def test_fused_step_capacity_mode_as_benchmarked_at_metric_size(gpu):
    """What bench.py times is FusedStep.run -> dgs_forward(capacity): the duplicate arrays are sized ahead, the count
    stays on the device (n_dev paths of the sort / ranges kernels).  At the metric size (37 M duplicates) that path must
    give, bit for bit, what the exact two-phase forward gives, and both must equal the autograd path
    (render_subframes + loss.backward) that test_config_as_benchmarked holds against the oracle -- same cameras, same
    upstream gradient -- so the oracle check covers the benchmarked kernels transitively."""
    import torch
    from deblurgs_amd import gaussian_renderer
    from deblurgs_amd.cloud import GaussianCloud
    from deblurgs_amd.fused_step import FusedStep
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    sc = synthetic.make_config("metric")
    K, P, W, H = sc["K"], sc["P"], sc["W"], sc["H"]
    C = synthetic.CONFIGS["metric"]["C"]
    dev = "cuda"
    cloud = GaussianCloud.from_scene(sc, dev)
    ref = RefCamera(W, H, sc["FoVx"], sc["FoVy"], device=dev)
    gt = torch.rand((1, 3, H, W), generator=torch.Generator().manual_seed(1234)).to(dev)
    m = CameraMotionModule(ref, gt, curve_order=C, num_subframes=K, device=dev)
    traj = synthetic.make_trajectory(K, C, sc["projection_matrix"], seed=0)
    with torch.no_grad():
        m._trans._control_points.copy_(torch.from_numpy(traj["ctrl_trans"])[None].to(dev))
        m._rot._control_points.copy_(torch.from_numpy(traj["ctrl_rot"])[None].to(dev))
    m.link_gaussian(cloud)
    assert m.is_optimizing()                        # the benchmarked step optimises the curves too
    bg = torch.tensor([0.3, 0.6, 0.1], device=dev)
    fs = FusedStep(cloud, m, lambda_hinge=0.0, speculative=True)
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]

    def snapshot(fr):
        torch.cuda.synchronize()
        g = {n: p.grad.clone() for n, p in zip(names, cloud.hot_parameters())}
        g.update(subframes=fr["subframes"].clone(), viewspace=fr["viewspace_grad"].clone(), radii=fr["radii"].clone(),
                 losses=fr["losses"].clone(), ct=m._trans._control_points.grad.clone(),
                 cr=m._rot._control_points.grad.clone())
        return g

    a_fr = fs.run(0, 1e-3, m.get_gt_image(0), bg)                      # exact two-phase forward: learns the count
    assert a_fr["skip_flag_ptr"] is None
    a = snapshot(a_fr)
    dsub, view, full, campos = (fs._keep[i].clone() for i in (6, 7, 8, 9))
    fs._poll(block=True)
    R = fs._seen[(0, K, 0)][-1]
    assert R > 30_000_000
    b_fr = fs.run(0, 1e-3, m.get_gt_image(0), bg)                      # capacity mode, as benchmarked
    assert b_fr["skip_flag_ptr"] is not None and fs.last_capacity == R + R // 4 + 16384
    b = snapshot(b_fr)
    fs._poll(block=True)
    assert fs.dropped == 0 and fs._seen[(0, K, 0)][-1] == R
    for key in a:
        assert torch.equal(a[key], b[key]), f"capacity mode vs exact forward: {key}"
    del a_fr, b_fr, fs
    torch.cuda.empty_cache()
    # the autograd path on the same cameras with the same upstream gradient
    for p in cloud.hot_parameters():
        p.grad = None
    view.requires_grad_(True)
    full.requires_grad_(True)
    pkg = gaussian_renderer.render_subframes(view, full, campos, ref, cloud, bg)
    assert torch.equal(pkg["render"], a["subframes"]) and torch.equal(pkg["radii"], a["radii"])
    (pkg["render"] * dsub).sum().backward()
    torch.cuda.synchronize()
    for n, p in zip(names, cloud.hot_parameters()):
        assert torch.equal(p.grad, b[n]), f"fused step vs autograd path: {n}"
    assert torch.equal(pkg["viewspace_points"].grad, b["viewspace"])
