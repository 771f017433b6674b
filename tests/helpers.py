"""Shared helpers of the parity tests: run the HIP path (through the C ABI, via the Python operator layer) and
the CPU oracle on the same seeded synthetic scene, and expose every intermediate of both for comparison."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from deblurgs_amd import synthetic  # noqa: E402
from oracle import oracle  # noqa: E402


def oracle_forward(scene, k, sh_degree=None, use_sigmoid=False, colors_precomp=None, cov3D_precomp=None,
                   scale_modifier=1.0, render=True):
    kw = dict(sh_degree=scene["sh_degree"] if sh_degree is None else sh_degree, use_sigmoid=use_sigmoid,
              scale_modifier=scale_modifier, z_far=scene["z_far"], render=render)
    if colors_precomp is None:
        kw["sh"] = scene["sh"]
    else:
        kw["colors_precomp"] = colors_precomp
    if cov3D_precomp is None:
        kw["scales"] = scene["scales"]
        kw["rotations"] = scene["rotations"]
    else:
        kw["cov3D_precomp"] = cov3D_precomp
    return oracle.forward(scene["means3D"], scene["opacities"], scene["viewmatrix"][k], scene["projmatrix"][k],
                          scene["campos"][k], scene["bg"], scene["W"], scene["H"], scene["tanfovx"], scene["tanfovy"],
                          **kw)


def unstable_pixels(st, alpha_tol=5e-7, power_tol=1e-4, T_tol=1e-8):
    """Pixels where some (pixel, Gaussian) pair of the ORACLE's traversal sits within a small margin of one of the
    reference's three thresholds (power > 0, alpha < 1/255, T(1-alpha) < 1e-4).  exp() differs by an ulp or
    two between glibc, CUDA libdevice and the gfx950 hardware exp, so such a pair may legitimately fall on
    either side; everywhere else the images must agree to 1e-4."""
    W, H = st["W"], st["H"]
    gx = (W + 15) // 16
    means2D, co = st["means2D"], st["conic_opacity"]
    flag = np.zeros(W * H, bool)
    ranges, pl = st["ranges"], st["point_list"]
    for tile in range(ranges.shape[0]):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        if r1 <= r0:
            continue
        ty, tx = divmod(tile, gx)
        ys, xs = np.meshgrid(np.arange(ty * 16, min(ty * 16 + 16, H)), np.arange(tx * 16, min(tx * 16 + 16, W)),
                             indexing="ij")
        px, py = xs.reshape(-1).astype(np.float32), ys.reshape(-1).astype(np.float32)
        g = pl[r0:r1]
        dx = means2D[g, 0][None] - px[:, None]
        dy = means2D[g, 1][None] - py[:, None]
        c = co[g]
        power = (-0.5 * (c[:, 0][None] * dx * dx + c[:, 2][None] * dy * dy) - c[:, 1][None] * dx * dy).astype(
            np.float32)
        alpha = np.minimum(0.99, c[:, 3][None] * np.exp(power)).astype(np.float32)
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)
        a = np.where(valid, alpha, 0.0)
        T_incl = np.cumprod(1.0 - a, axis=1)
        stop = valid & (T_incl < 1e-4)
        reached = (np.cumsum(stop, axis=1) - stop) == 0       # pairs the pixel actually evaluates
        near = (np.abs(alpha - 1.0 / 255.0) < alpha_tol) | (np.abs(power) < power_tol)
        near |= valid & (np.abs(T_incl - 1e-4) < T_tol)
        near &= reached
        flag[(ys.reshape(-1) * W + xs.reshape(-1))[near.any(axis=1)]] = True
    return flag.reshape(H, W)


# --------------------------------------------------------------------------------------------- HIP side
def _t(a, device="cuda"):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def hip_settings(scene, K, sh_degree=None, use_sigmoid=False, scale_modifier=1.0, debug=False, campos=None):
    from deblurgs_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    cam = scene["campos"][:K] if campos is None else campos
    return GaussianRasterizationSettings(
        image_height=scene["H"], image_width=scene["W"], tanfovx=scene["tanfovx"], tanfovy=scene["tanfovy"],
        bg=_t(scene["bg"]), scale_modifier=scale_modifier, z_near=scene["z_near"], z_far=scene["z_far"],
        use_sigmoid=use_sigmoid, sh_degree=scene["sh_degree"] if sh_degree is None else sh_degree,
        campos=_t(cam if K > 1 else cam[0]), prefiltered=False, debug=debug)


import contextlib


@contextlib.contextmanager
def tile_cull(flag):
    """Selects the duplicate rule of the HIP path for the calls inside: False = the reference's rectangle lists."""
    from deblurgs_amd import diff_gaussian_rasterization as dgr
    old = dgr.TILE_CULL
    dgr.TILE_CULL = bool(flag)
    try:
        yield
    finally:
        dgr.TILE_CULL = old


def hip_forward_state(scene, K, sh_degree=None, use_sigmoid=False, colors_precomp=None, cov3D_precomp=None,
                      scale_modifier=1.0, cull=False):
    """Runs the fused forward through the C ABI and returns outputs + every saved sub-array as numpy.  cull=False
    (default) keeps the reference's duplicate lists so that keys / point_list / ranges compare bit for bit; with
    cull=True the low key word is the duplicate's emission index instead of the depth bits."""
    with tile_cull(cull):
        return _hip_forward_state(scene, K, sh_degree, use_sigmoid, colors_precomp, cov3D_precomp, scale_modifier)


def _hip_forward_state(scene, K, sh_degree, use_sigmoid, colors_precomp, cov3D_precomp, scale_modifier):
    import torch
    from deblurgs_amd import _lib
    from deblurgs_amd import diff_gaussian_rasterization as dgr
    rs = hip_settings(scene, K, sh_degree, use_sigmoid, scale_modifier)
    rs = rs._replace(campos=_t(scene["campos"][:K]))
    sh = None if colors_precomp is not None else _t(scene["sh"])
    col = None if colors_precomp is None else _t(colors_precomp)
    sc = None if cov3D_precomp is not None else _t(scene["scales"])
    rot = None if cov3D_precomp is not None else _t(scene["rotations"])
    cov = None if cov3D_precomp is None else _t(cov3D_precomp)
    R, color, depth, radii, geom, binning, image = dgr._forward_impl(
        K, _t(scene["means3D"]), sh, col, _t(scene["opacities"]).reshape(-1), sc, rot, cov,
        _t(scene["viewmatrix"][:K]), _t(scene["projmatrix"][:K]), _t(scene["campos"][:K]), rs)
    torch.cuda.synchronize()
    P, W, H = scene["P"], scene["W"], scene["H"]
    N = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)
    L = _lib.layout(P, W, H, K, R)

    def view(blob, off, nbytes, dtype, shape):
        return blob[off:off + nbytes].cpu().numpy().view(dtype).reshape(shape)

    rows = view(geom, L.geom_rows, K * P * 48, np.float32, (K, P, 12))
    st = dict(
        R=R, K=K, color=color.cpu().numpy(), depth=depth.cpu().numpy(), radii=radii.cpu().numpy(),
        rows=rows, rows_u32=rows.view(np.uint32),
        cov3D=view(geom, L.cov3D, P * 24, np.float32, (P, 6)),
        pre_sigmoid=view(geom, L.pre_sigmoid, K * P * 12, np.float32, (K, P, 3)),
        tiles_touched=view(geom, L.tiles_touched, K * P * 4, np.uint32, (K, P)),
        point_offsets=view(geom, L.point_offsets, K * P * 4, np.uint32, (K, P)),
        tt_tight=view(geom, L.tt_tight, K * P * 4, np.uint32, (K * P,)),
        offs_tight=view(geom, L.offs_tight, K * P * 4, np.uint32, (K * P,)),
        final_T=view(image, L.final_T, K * N * 4, np.float32, (K, N)),
        n_contrib=view(image, L.n_contrib, K * N * 4, np.uint32, (K, N)),
        ranges=view(image, L.ranges, K * T * 8, np.uint32, (K, T, 2)),
        keys=view(binning, L.keys_sorted, R * 8, np.uint64, (R,)),
        point_list=view(binning, L.point_list, R * 4, np.uint32, (R,)),
        sort_bits=L.sort_bits, T=T,
    )
    return st


def hip_forward_backward(scene, K, dL_dcolor, dL_ddepth=None, sh_degree=None, use_sigmoid=False, fused=True,
                         colors_precomp=None, cov3D_precomp=None):
    """Forward + backward through the public operator (autograd).  fused=False uses the K=1 reference API K
    times (and sums per-Gaussian grads like autograd does in the reference loop)."""
    import torch
    from deblurgs_amd.diff_gaussian_rasterization import GaussianRasterizer
    dev = "cuda"
    names = ["means3D", "opacities", "sh", "scales", "rotations"]
    inp = {n: _t(scene[n]).requires_grad_(True) for n in names}
    col = None if colors_precomp is None else _t(colors_precomp).requires_grad_(True)
    cov = None if cov3D_precomp is None else _t(cov3D_precomp).requires_grad_(True)
    view = _t(scene["viewmatrix"][:K]).requires_grad_(True)
    proj = _t(scene["projmatrix"][:K]).requires_grad_(True)
    gC = _t(dL_dcolor)
    gD = None if dL_ddepth is None else _t(dL_ddepth)
    P = scene["P"]
    kw = dict(shs=inp["sh"] if col is None else None, colors_precomp=col,
              scales=inp["scales"] if cov is None else None, rotations=inp["rotations"] if cov is None else None,
              cov3D_precomp=cov)
    if fused:
        rs = hip_settings(scene, K, sh_degree, use_sigmoid)._replace(campos=_t(scene["campos"][:K]))
        m2 = torch.zeros((K, P, 3), device=dev, requires_grad=True)
        color, depth, radii = GaussianRasterizer(rs).forward_subframes(
            inp["means3D"], m2, inp["opacities"], viewmatrices=view, projmatrices=proj, **kw)
        loss = (color * gC).sum()
        if gD is not None:
            loss = loss + (depth * gD).sum()
        loss.backward()
        m2g = m2.grad
    else:
        colors, depths, radiis, m2s = [], [], [], []
        loss = 0
        for k in range(K):
            rs = hip_settings(scene, 1, sh_degree, use_sigmoid, campos=scene["campos"][k:k + 1])
            m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
            c, d, r = GaussianRasterizer(rs)(inp["means3D"], m2, inp["opacities"], viewmatrix=view[k],
                                             projmatrix=proj[k], **kw)
            loss = loss + (c * gC[k]).sum()
            if gD is not None:
                loss = loss + (d * gD[k]).sum()
            colors.append(c)
            depths.append(d)
            radiis.append(r)
            m2s.append(m2)
        loss.backward()
        color, depth, radii = torch.stack(colors), torch.stack(depths), torch.stack(radiis)
        m2g = torch.stack([m.grad for m in m2s])
    torch.cuda.synchronize()
    out = dict(color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(), radii=radii.cpu().numpy(),
               dL_dmeans2D=m2g.cpu().numpy(), dL_dviewmatrix=view.grad.cpu().numpy(),
               dL_dprojmatrix=proj.grad.cpu().numpy())
    for n in names:
        out["dL_d" + n] = None if inp[n].grad is None else inp[n].grad.cpu().numpy()
    out["dL_dcolors_precomp"] = None if col is None or col.grad is None else col.grad.cpu().numpy()
    out["dL_dcov3D_precomp"] = None if cov is None or cov.grad is None else cov.grad.cpu().numpy()
    return out


def oracle_forward_backward(scene, K, dL_dcolor, dL_ddepth=None, sh_degree=None, use_sigmoid=False,
                            colors_precomp=None, cov3D_precomp=None):
    """Reference semantics: K independent renders, per-Gaussian grads summed over k (autograd accumulation)."""
    outs = []
    for k in range(K):
        st = oracle_forward(scene, k, sh_degree, use_sigmoid, colors_precomp, cov3D_precomp)
        g = oracle.backward(st, dL_dcolor[k], None if dL_ddepth is None else dL_ddepth[k])
        outs.append((st, g))
    res = dict(
        color=np.stack([o[0]["color"] for o in outs]), depth=np.stack([o[0]["depth"] for o in outs]),
        radii=np.stack([o[0]["radii"] for o in outs]),
        dL_dmeans2D=np.stack([o[1]["dL_dmeans2D"] for o in outs]),
        dL_dviewmatrix=np.stack([o[1]["dL_dviewmatrix"] for o in outs]),
        dL_dprojmatrix=np.stack([o[1]["dL_dprojmatrix"] for o in outs]),
        dL_dmeans3D=sum(o[1]["dL_dmeans3D"] for o in outs), dL_dopacities=sum(o[1]["dL_dopacity"] for o in outs),
        dL_dsh=sum(o[1]["dL_dsh"] for o in outs), dL_dscales=sum(o[1]["dL_dscales"] for o in outs),
        dL_drotations=sum(o[1]["dL_drotations"] for o in outs),
        dL_dcolors_precomp=sum(o[1]["dL_dcolors"] for o in outs),
        dL_dcov3D_precomp=sum(o[1]["dL_dcov3D"] for o in outs),
        states=[o[0] for o in outs],
    )
    return res


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
