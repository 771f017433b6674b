"""Naive dense differentiable PyTorch (CPU) Gaussian-splat rasteriser.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  Two jobs:
  1. autograd ground truth (float64-capable) for every gradient the reference's backward.cu derives
     analytically: d/d{means3D, SH, opacity, scale, rotation}, and -- because the rotation block ``W`` of the
     view matrix is detached inside the 2-D covariance exactly like the reference omits it
     (backward.cu:277-294) -- also the reference's dL_dviewmatrix.  dL_dprojmatrix is checked on the entries
     for which the reference's formula is a scaled analytic gradient (backward.cu:430-450).
  2. the "naive PyTorch CPU rasteriser" that BASELINE.json's config 1 names, timed by bench.py's
     cpu_baseline leg on a bounded sample.

Semantics follow forward.cu:166-392: near cull at z<=0.2, EWA covariance with +0.3 low-pass, tile-rectangle
membership from ceil(3*sqrt(lambda_max)), per-pixel front-to-back compositing in (depth, index) order with the
alpha>=1/255 / power<=0 / T*(1-alpha)<1e-4 rules, colour + depth outputs with background / z_far terms.
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def eval_sh_color(deg, sh, dirs):
    """sh [P,M,3], dirs [P,3] unit -> [P,3] (forward.cu:20-61, before the activation)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    result = SH_C0 * sh[:, 0]
    if deg > 0:
        result = result - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            result = (result + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                      + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                      + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                result = (result + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                          + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                          + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                          + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                          + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return result


def quat_to_rotmat(q):
    """Un-normalised quaternion (r,x,y,z) -> the reference's R (forward.cu:138-148, row-major here)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, sh=None,
               colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
               scale_modifier=1.0, use_sigmoid=False, pix_offset=None):
    dt = means3D.dtype
    P = means3D.shape[0]
    V = viewmatrix.reshape(4, 4)   # row-vector convention: p_view = [p,1] @ V
    F = projmatrix.reshape(4, 4)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    p_view = ph @ V[:, :3]
    p_hom = ph @ F
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w[:, None]
    depth = p_view[:, 2]
    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            1).reshape(P, 3, 3)
    else:
        R = quat_to_rotmat(rotations)
        S = scale_modifier * scales
        RS = R * S[:, None, :]      # R @ diag(S)
        Sigma = RS @ RS.transpose(1, 2)
    focal_x = W / (2.0 * tanfovx)
    focal_y = H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tx = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    Jm = torch.stack([focal_x / tz, zero, -(focal_x * tx) / (tz * tz),
                      zero, focal_y / tz, -(focal_y * ty) / (tz * tz)], 1).reshape(P, 2, 3)
    Rw = V[:3, :3].detach().T      # world->camera rotation; NOT differentiated in the reference
    A = Jm @ Rw
    cov = A @ Sigma @ A.transpose(1, 2)
    ca = cov[:, 0, 0] + 0.3
    cb = cov[:, 0, 1]
    cc = cov[:, 1, 1] + 0.3
    det = ca * cc - cb * cb
    det_inv = 1.0 / det
    conic = torch.stack([cc * det_inv, -cb * det_inv, ca * det_inv], 1)
    mid = 0.5 * (ca + cc)
    disc = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + disc, mid - disc))).detach()
    pix = torch.stack([((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5, ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    if pix_offset is not None:     # a zero [P,2] leaf: its .grad is dL/d(pixel-space mean), the reference's means2D carrier
        pix = pix + pix_offset     # before the 0.5 W / 0.5 H NDC scaling of backward.cu:535-536
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pd, rd = pix.detach(), radius
    minx = torch.clamp(torch.trunc((pd[:, 0] - rd) / 16), 0, gx)
    miny = torch.clamp(torch.trunc((pd[:, 1] - rd) / 16), 0, gy)
    maxx = torch.clamp(torch.trunc((pd[:, 0] + rd + 15) / 16), 0, gx)
    maxy = torch.clamp(torch.trunc((pd[:, 1] + rd + 15) / 16), 0, gy)
    visible = (depth.detach() > 0.2) & (det.detach() != 0) & ((maxx - minx) * (maxy - miny) > 0)
    if colors_precomp is not None:
        color = colors_precomp
    else:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        raw = eval_sh_color(sh_degree, sh, d)
        color = torch.sigmoid(raw) if use_sigmoid else torch.clamp(raw + 0.5, min=0.0)
    return dict(depth=depth, pix=pix, conic=conic, opacity=opacities.reshape(-1), color=color, radius=radius,
                rect=(minx, miny, maxx, maxy), visible=visible)


def rasterize(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, z_far=100.0,
              pixel_chunk=8192, window=None, **kw):
    """Returns (color [3,H,W], depth [1,H,W], radii [P]).  All tensor inputs may require grad.
    window=(x0, y0, x1, y1): render only that pixel rectangle of the W x H image (outputs [3,y1-y0,x1-x0] /
    [1,y1-y0,x1-x0]) from the Gaussians whose tile rectangle meets it -- the bounded sample bench.py times."""
    g = preprocess(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, **kw)
    dt = means3D.dtype
    vis = g["visible"]
    idx = torch.nonzero(vis)[:, 0]
    # (depth, index) order == the stable radix sort of (tile|depth) keys restricted to one tile
    order = torch.argsort(g["depth"].detach()[idx], stable=True)
    idx = idx[order]
    pix, conic, op, col, dep = g["pix"][idx], g["conic"][idx], g["opacity"][idx], g["color"][idx], g["depth"][idx]
    minx, miny, maxx, maxy = (r[idx] for r in g["rect"])
    if window is not None:
        x0, y0, x1, y1 = window
        meets = (minx * 16 < x1) & (maxx * 16 > x0) & (miny * 16 < y1) & (maxy * 16 > y0)
        pix, conic, op, col, dep = pix[meets], conic[meets], op[meets], col[meets], dep[meets]
        minx, miny, maxx, maxy = minx[meets], miny[meets], maxx[meets], maxy[meets]
        wpix = torch.stack(torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij"), -1).reshape(-1, 2)
        all_pid = wpix[:, 0] * W + wpix[:, 1]
    else:
        x0, y0, x1, y1 = 0, 0, W, H
        all_pid = torch.arange(W * H)
    N = all_pid.shape[0]
    out_c = []
    out_d = []
    for s in range(0, N, pixel_chunk):
        pid = all_pid[s:min(s + pixel_chunk, N)]
        pxi, pyi = pid % W, pid // W
        px, py = pxi.to(dt), pyi.to(dt)
        tx_, ty_ = (pxi // 16).to(dt), (pyi // 16).to(dt)
        in_rect = ((tx_[:, None] >= minx[None]) & (tx_[:, None] < maxx[None])
                   & (ty_[:, None] >= miny[None]) & (ty_[:, None] < maxy[None]))
        dx = pix[None, :, 0] - px[:, None]
        dy = pix[None, :, 1] - py[:, None]
        power = -0.5 * (conic[None, :, 0] * dx * dx + conic[None, :, 2] * dy * dy) - conic[None, :, 1] * dx * dy
        raw_alpha = op[None] * torch.exp(power)
        # The reference's backward ignores the min(0.99, .) clamp (backward.cu:576-637 differentiates
        # alpha = opacity*G unconditionally), i.e. a straight-through clamp.
        alpha = raw_alpha + (torch.clamp(raw_alpha, max=0.99) - raw_alpha).detach()
        valid = in_rect & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
        alpha = torch.where(valid, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - alpha
        T_incl = torch.cumprod(one_m, dim=1)
        T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
        # done as soon as a valid Gaussian would push T below 1e-4; that Gaussian is NOT blended
        stop = (valid & (T_incl.detach() < 0.0001)).to(torch.int32)
        alive = torch.cumsum(stop, dim=1) == 0
        w = torch.where(alive, alpha * T_excl, torch.zeros_like(alpha))
        T_final = torch.where(alive, one_m, torch.ones_like(one_m)).prod(dim=1)
        C = w @ col + T_final[:, None] * bg[None]
        Dp = w @ dep + T_final * z_far
        out_c.append(C)
        out_d.append(Dp)
    color = torch.cat(out_c, 0).T.reshape(3, y1 - y0, x1 - x0)
    depth = torch.cat(out_d, 0).reshape(1, y1 - y0, x1 - x0)
    radii = torch.where(vis, g["radius"], torch.zeros_like(g["radius"])).to(torch.int32)
    return color, depth, radii
