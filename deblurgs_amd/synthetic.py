"""Deterministic synthetic Gaussian clouds and SE(3)-Bezier trajectories (BASELINE.md
section 3).  There is no dataset on the GPU box, so bench.py, the parity tests and smoke() all draw their
inputs from here.  Pure numpy: the same bytes feed the CPU oracle and the HIP path.
"""
import math

import numpy as np

SH_C0 = 0.28209479177387814

# BASELINE.json configs (P, W, H, K, curve_order); "metric" is the configuration the headline metric is quoted on.
CONFIGS = {
    "cfg1": dict(P=1_000, W=256, H=256, K=1, C=3),
    "cfg2": dict(P=100_000, W=800, H=800, K=9, C=9),
    "cfg3": dict(P=1_000_000, W=1600, H=1200, K=15, C=3),
    "metric": dict(P=1_000_000, W=1920, H=1080, K=15, C=3),
    "cfg5": dict(P=5_000_000, W=3840, H=2160, K=31, C=5),
}


def projection_matrix(znear, zfar, fovx, fovy):
    """utils/graphics_utils.py:51-71 in float32 steps (torch.zeros(4,4) is float32 there)."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def _hat(v):
    x, y, z = v
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], np.float64)


def se3_exp_np(log_transform, eps=1e-4):
    """float64 numpy restatement of utils/pytorch3d_functions.py:373-457 for one 6-vector -> 4x4 (row-vector form)."""
    lt, lr = np.asarray(log_transform[:3], np.float64), np.asarray(log_transform[3:], np.float64)
    ang = math.sqrt(max(float(lr @ lr), eps))
    S = _hat(lr)
    S2 = S @ S
    R = (math.sin(ang) / ang) * S + ((1 - math.cos(ang)) / ang ** 2) * S2 + np.eye(3)
    V = np.eye(3) + S * ((1 - math.cos(ang)) / ang ** 2) + S2 * ((ang - math.sin(ang)) / ang ** 3)
    out = np.zeros((4, 4), np.float64)
    out[:3, :3] = R
    out[:3, 3] = V @ lt
    out[3, 3] = 1.0
    return out.T


def bezier_np(ctrl, t):
    """scene/bezier.py:54-83: ctrl [C+1,d], t [K] -> [K,d]; control point 0 is reached at t = 1."""
    C = ctrl.shape[0] - 1
    k = np.arange(C + 1)
    coeff = (t[:, None] ** (C - k)[None]) * ((1 - t)[:, None] ** k[None]) * np.array([math.comb(C, i) for i in k])
    return coeff @ ctrl


def make_camera(W, H, fovx_deg=60.0, znear=0.01, zfar=100.0):
    tanfovx = math.tan(math.radians(fovx_deg) * 0.5)
    tanfovy = tanfovx * H / W
    fovx = 2 * math.atan(tanfovx)
    fovy = 2 * math.atan(tanfovy)
    proj = projection_matrix(znear, zfar, fovx, fovy).T.copy()  # stored transposed like scene/cameras.py:58
    return dict(W=W, H=H, tanfovx=tanfovx, tanfovy=tanfovy, FoVx=fovx, FoVy=fovy, znear=znear, zfar=zfar,
                projection_matrix=proj)


def make_trajectory(K, curve_order, proj_T, seed=0, trans_sigma=0.01, rot_sigma=0.002):
    """K subframe cameras along an SE(3) Bezier around the identity pose (scene/motion.py:248-294)."""
    rng = np.random.default_rng(seed + 1000)
    ctrl_t = rng.normal(0.0, trans_sigma, (curve_order + 1, 3))
    ctrl_r = rng.normal(0.0, rot_sigma, (curve_order + 1, 3))
    nu = np.linspace(0.0, 1.0, K) if K > 1 else np.zeros(1)
    se3 = np.concatenate([bezier_np(ctrl_t, nu), bezier_np(ctrl_r, nu)], axis=1)
    view = np.zeros((K, 4, 4), np.float32)
    full = np.zeros((K, 4, 4), np.float32)
    campos = np.zeros((K, 3), np.float32)
    for k in range(K):
        c2w = se3_exp_np(se3[k])
        rot = c2w[:3, :3].T
        trans = c2w[3, :3]
        wv = np.eye(4, dtype=np.float32)
        wv[:3, :3] = rot.astype(np.float32)
        wv[3, :3] = (-trans @ rot).astype(np.float32)
        view[k] = wv
        full[k] = wv @ proj_T.astype(np.float32)
        campos[k] = np.linalg.inv(wv)[3, :3]
    return dict(viewmatrix=view, projmatrix=full, campos=campos, se3=se3.astype(np.float32), nu=nu.astype(np.float32),
                ctrl_trans=ctrl_t.astype(np.float32), ctrl_rot=ctrl_r.astype(np.float32))


def make_scene(P, W, H, K=1, curve_order=3, sh_degree=2, seed=0, sigma_px=1.5, sigma_log=0.8, max_sh_degree=None):
    """Synthetic cloud + camera + trajectory exactly as SURVEY.md section 8d prescribes."""
    rng = np.random.default_rng(seed)
    cam = make_camera(W, H)
    tanfovx, tanfovy = cam["tanfovx"], cam["tanfovy"]
    focal = W / (2.0 * tanfovx)
    z = rng.uniform(1.0, 10.0, P)
    u = rng.uniform(-1.15, 1.15, P)
    v = rng.uniform(-1.15, 1.15, P)
    means3D = np.stack([u * tanfovx * z, v * tanfovy * z, z], axis=1).astype(np.float32)
    sig = np.exp(rng.normal(math.log(sigma_px), sigma_log, (P, 3)))
    scales = (z[:, None] * sig / focal).astype(np.float32)
    q = rng.normal(0.0, 1.0, (P, 4))
    rotations = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opacities = rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32)
    Dmax = sh_degree if max_sh_degree is None else max_sh_degree
    M = (Dmax + 1) ** 2
    sh = np.zeros((P, M, 3), np.float32)
    sh[:, 0, :] = rng.normal(0.0, 1.0, (P, 3)) / SH_C0 * 0.3
    if M > 1:
        sh[:, 1:, :] = rng.normal(0.0, 0.05, (P, M - 1, 3))
    bg = rng.random(3).astype(np.float32)
    traj = make_trajectory(K, curve_order, cam["projection_matrix"], seed=seed)
    scene = dict(P=P, W=W, H=H, K=K, sh_degree=sh_degree, M=M, means3D=means3D, scales=scales, rotations=rotations,
                 opacities=opacities, sh=sh, bg=bg, z_near=0.2, z_far=100.0, scale_modifier=1.0)
    scene.update(cam)
    scene.update(traj)
    return scene


def make_config(name, seed=0, **over):
    cfg = dict(CONFIGS[name])
    cfg.update(over)
    extra = {k: over[k] for k in ("sigma_px", "sigma_log") if k in over}
    return make_scene(cfg["P"], cfg["W"], cfg["H"], K=cfg["K"], curve_order=cfg["C"], seed=seed,
                      sh_degree=over.get("sh_degree", 2), **extra)
