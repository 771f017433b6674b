"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the per-Gaussian training-side step that follows the hot path
(SURVEY.md 8f, f3).  Nothing under deblurgs_amd/ may import this file.

  adam_step          torch/optim/adam.py (_single_tensor_adam, torch 2.10) as used by the reference with
                     Adam(lr=0.0, eps=1e-15) (scene/gaussian_model.py:195) -- float32 array ops, bias corrections
                     in python floats.
  densify_and_prune  scene/gaussian_model.py:436-448 with densify_and_clone :419-434, densify_and_split :389-417,
                     densification_postfix :366-387 (cat, zero moments), prune_points :336-349, written with the
                     reference's boolean masks and concatenations (the HIP path uses scans instead).
Pinned by tests/golden/densify_golden.npz, which the reference's own GaussianModel produced (make_golden_densify.py).
"""
import numpy as np

FIELDS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
f32 = np.float32


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-15):
    p, g, m, v = (np.asarray(a, f32) for a in (p, g, m, v))
    m = m + f32(1 - beta1) * (g - m)                       # lerp_
    v = v * f32(beta2) + (f32(1 - beta2) * g) * g          # mul_, addcmul_
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / f32(bc2 ** 0.5) + f32(eps)
    p = p + f32(-(lr / bc1)) * (m / denom)                 # addcdiv_
    return p.astype(f32), m.astype(f32), v.astype(f32)


def build_rotation(r):
    """utils/general_utils.py:117-138."""
    q = r / np.sqrt((r * r).sum(1, dtype=f32))[:, None]
    R = np.zeros((q.shape[0], 3, 3), f32)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_prune(params, m, v, accum, denom, max_grad, extent, percent_dense, noise, scale_lb=0.0, alpha_lb=0.0,
                      isotropic=False):
    """params / m / v: dicts over FIELDS (m / v may be None = no optimiser state).  noise: [2 m_sel, 3] unit normals in
    the reference's draw order (copy-major).  Returns (params, m, v) of the new cloud."""
    params = {k: np.asarray(a, f32) for k, a in params.items()}
    has = m is not None
    m = {k: np.asarray(a, f32) for k, a in m.items()} if has else {k: np.zeros_like(a) for k, a in params.items()}
    v = {k: np.asarray(a, f32) for k, a in v.items()} if has else {k: np.zeros_like(a) for k, a in params.items()}
    # get_scaling (scene/gaussian_model.py:114-119): column 0 expanded to three when use_isotrophic
    raw_scaling = lambda: np.repeat(params["scaling"][:, :1], 3, 1) if isotropic else params["scaling"]
    get_scaling = lambda: np.exp(raw_scaling()) + f32(scale_lb)
    with np.errstate(all="ignore"):
        grads = (np.asarray(accum, f32) / np.asarray(denom, f32)).reshape(-1)
    grads[np.isnan(grads)] = 0.0

    def cat(new):
        for k in FIELDS:
            params[k] = np.concatenate([params[k], new[k]], 0)
            m[k] = np.concatenate([m[k], np.zeros_like(new[k])], 0)
            v[k] = np.concatenate([v[k], np.zeros_like(new[k])], 0)

    def prune(mask):
        for k in FIELDS:
            params[k], m[k], v[k] = params[k][~mask], m[k][~mask], v[k][~mask]

    # densify_and_clone
    sel = (np.abs(grads) >= f32(max_grad)) & (get_scaling().max(1) <= f32(percent_dense * extent))
    cat({k: params[k][sel] for k in FIELDS})
    # densify_and_split (N = 2)
    n_init = params["xyz"].shape[0]
    padded = np.zeros(n_init, f32)
    padded[:grads.shape[0]] = grads
    sel = (padded >= f32(max_grad)) & (get_scaling().max(1) > f32(percent_dense * extent))
    stds = np.tile(get_scaling()[sel], (2, 1))
    samples = stds * np.asarray(noise, f32).reshape(-1, 3)
    rots = np.tile(build_rotation(params["rotation"][sel]), (2, 1, 1))
    new = {k: np.tile(params[k][sel], (2,) + (1,) * (params[k].ndim - 1)) for k in FIELDS}
    new["xyz"] = np.einsum("nij,nj->ni", rots, samples).astype(f32) + new["xyz"]
    new["scaling"] = np.log(np.maximum(stds / f32(0.8 * 2) - f32(scale_lb), f32(0.001))).astype(f32)
    cat(new)
    prune(np.concatenate([sel, np.zeros(2 * int(sel.sum()), bool)]))
    # opacity prune
    min_opacity = alpha_lb + (1 - alpha_lb) * 0.005
    prune((np.clip(params["opacity"], 0.0, 1.0) < f32(min_opacity)).reshape(-1))
    return params, (m if has else None), (v if has else None)
