"""CPU tests: the oracle and the product's host-side code against the golden vectors captured from the
reference's importable Python (tests/golden/make_golden.py), plus the oracle's backward against float64
autograd of the naive torch rasteriser."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import oracle_forward, synthetic
from oracle import oracle, torch_naive

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_pose_path_matches_reference_se3_exp_map():
    from deblurgs_amd import pose
    g = load("pose_golden.npz")
    se3 = torch.tensor(g["se3"])
    assert np.abs(pose.se3_exp_map(se3).numpy() - g["exp32"]).max() <= 2e-6
    assert np.abs(pose.se3_exp_map(se3.double()).numpy() - g["exp64"]).max() <= 1e-12
    assert np.abs(pose.so3_exp_map(se3[:, 3:]).numpy() - g["so3_exp32"]).max() <= 2e-6
    # the numpy generator used by bench/tests follows the same map (row-vector 4x4)
    for i in range(se3.shape[0]):
        assert np.abs(synthetic.se3_exp_np(g["se3"][i].astype(np.float64)) - g["exp64"][i]).max() <= 1e-6
    # exp is a right inverse of the reference's log on these inputs
    back = pose.se3_exp_map(torch.tensor(g["log_of_exp64"]))
    assert np.abs(back.numpy() - g["exp64"]).max() <= 1e-9


def test_pose_oracle_is_pinned_and_checks_the_product_maps():
    """oracle/pose_oracle.py (the function-by-function restatement of utils/pytorch3d_functions.py, test infrastructure)
    against the reference-generated golden vectors, and the product's own formulation (deblurgs_amd/pose.py) against it on
    fresh inputs: exact zeros, tiny angles (the eps clamp), angles near pi."""
    from deblurgs_amd import pose
    from oracle import pose_oracle as po
    g = load("pose_golden.npz")
    se3 = torch.tensor(g["se3"])
    assert np.abs(po.se3_exp_map(se3.double()).numpy() - g["exp64"]).max() <= 1e-12
    assert np.abs(po.se3_exp_map(se3).numpy() - g["exp32"]).max() <= 2e-6
    assert np.abs(po.se3_log_map(torch.tensor(g["exp64"])).numpy() - g["log_of_exp64"]).max() <= 1e-12
    torch.manual_seed(11)
    x = torch.randn(200, 6, dtype=torch.float64)
    x[:20, 3:] *= 1e-5
    x[20] = 0.0
    x[21:40, 3:] *= 3.1 / x[21:40, 3:].norm(dim=1, keepdim=True)
    for dt, tol in ((torch.float64, 1e-13), (torch.float32, 1e-6)):
        T = po.se3_exp_map(x.to(dt))
        assert float((pose.se3_exp_map(x.to(dt)) - T).abs().max()) <= tol
        assert float((pose.so3_exp_map(x[:, 3:].to(dt)) - po.so3_exp_map(x[:, 3:].to(dt))).abs().max()) <= tol
        assert float((pose.se3_log_map(T) - po.se3_log_map(T)).abs().max()) <= tol * 10
        assert float((pose.so3_log_map(T[:, :3, :3]) - po.so3_log_map(T[:, :3, :3])).abs().max()) <= tol * 10


def test_se3_log_map_matches_reference():
    """pose.se3_log_map against the reference's own se3_log_map outputs (utils/pytorch3d_functions.py:462-540, golden
    `log_of_exp64`), including the exact-zero, near-zero (Taylor branch of phi / 2 sin phi) and large-angle rows."""
    from deblurgs_amd import pose
    g = load("pose_golden.npz")
    T64 = torch.tensor(g["exp64"])
    assert np.abs(pose.se3_log_map(T64).numpy() - g["log_of_exp64"]).max() <= 1e-12
    lg32 = pose.se3_log_map(torch.tensor(g["exp32"])).numpy()
    assert np.abs(lg32 - g["log_of_exp64"]).max() <= 2e-3           # acos near 1 in float32, as in the reference
    big = np.linalg.norm(g["se3"][:, 3:], axis=1) > 0.05
    assert np.abs(lg32[big] - g["log_of_exp64"][big]).max() <= 2e-5
    with pytest.raises(ValueError):
        bad = T64.clone()
        bad[0, 0, 3] = 0.5
        pose.se3_log_map(bad)


def test_motion_module_initialises_from_camera_poses_with_the_real_log():
    """scene/motion.py:196-205: the curve's control points start at se3_log_map of the dataset pose (far from the
    identity here), so that every subframe camera of the fresh module reproduces that pose."""
    from deblurgs_amd import pose
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    g = load("pose_golden.npz")
    torch.manual_seed(0)
    c2w = torch.tensor(g["exp64"]).float()                       # row-vector [[R,0],[T,1]]
    rot, trans = c2w[:, :3, :3].transpose(-2, -1), c2w[:, 3, :3]   # what the reference reads from CameraInfo
    ref = RefCamera(64, 48, 1.0, 0.8, device="cpu")
    m = CameraMotionModule(ref, torch.zeros(c2w.shape[0], 3, 4, 4), curve_order=3, num_subframes=5, device="cpu",
                           init_c2w=(rot, trans))
    want = torch.tensor(g["log_of_exp64"]).float()
    assert (m._rot._control_points - want[:, None, 3:]).abs().max() <= 6e-3       # + N(0, 0.001) initial noise
    assert (m._trans._control_points - want[:, None, :3]).abs().max() <= 6e-3
    for i in (2, 5, 11):
        wv, _, cc = m.get_trajectory_matrices(i, fused=False)
        assert (wv[:, :3, :3] - rot[i]).abs().max() <= 1e-2 and (cc - trans[i]).abs().max() <= 1e-2


def test_quaternion_curve_type_against_scipy():
    """curve_type="quarternion_cartesian" (scene/motion.py:191-194,242-246) without `roma`: the two conversions are the
    published SciPy algorithm roma adapts (XYZW, no sign canonicalisation), pinned here against scipy itself."""
    from scipy.spatial.transform import Rotation
    from deblurgs_amd import pose
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    Rm = Rotation.random(300, random_state=1).as_matrix()
    Rm[0] = np.eye(3)
    Rm[1] = Rotation.from_rotvec([np.pi - 1e-6, 0, 0]).as_matrix()          # trace ~ -1: pivot on a diagonal entry
    Rm[2] = Rotation.from_rotvec([0, np.pi * 0.9999, 0]).as_matrix()
    Rm[3] = Rotation.from_rotvec([0.3, 0.2, np.pi * 0.99]).as_matrix()
    q = pose.rotmat_to_unitquat(torch.tensor(Rm))
    assert np.abs(q.numpy() - Rotation.from_matrix(Rm).as_quat()).max() <= 1e-14
    assert np.abs(pose.unitquat_to_rotmat(q).numpy() - Rm).max() <= 1e-14
    assert np.abs(pose.unitquat_to_rotmat(q.float()).numpy() - Rotation.from_quat(q.numpy()).as_matrix()).max() <= 1e-6
    # the module: control points = the pose's quaternion (+ noise), positions in Cartesian space
    torch.manual_seed(1)
    rot, trans = torch.tensor(Rm[:6]).float(), torch.randn(6, 3)
    ref = RefCamera(64, 48, 1.0, 0.8, device="cpu")
    m = CameraMotionModule(ref, torch.zeros(6, 3, 4, 4), curve_order=4, num_subframes=7, device="cpu",
                           curve_type="quarternion_cartesian", init_c2w=(rot, trans))
    assert m._rot._control_points.shape == (6, 5, 4) and m._trans._control_points.shape == (6, 5, 3)
    with torch.no_grad():                                   # no noise: every sample must be the initial pose exactly
        m._rot._control_points.copy_(pose.rotmat_to_unitquat(rot)[:, None, :].expand(-1, 5, -1))
        m._trans._control_points.copy_(trans[:, None, :].expand(-1, 5, -1))
    r, t = m._sample_c2w_from_nu(3)
    assert (r - rot[3]).abs().max() <= 1e-6 and (t - trans[3]).abs().max() <= 1e-6
    wv, fp, cc = m.get_trajectory_matrices(3)
    assert (wv[:, :3, :3] - rot[3]).abs().max() <= 1e-6 and (cc - trans[3]).abs().max() <= 1e-5
    (wv.sum() + fp.sum()).backward()
    assert m._rot._control_points.grad.abs().sum() > 0 and m._nu.grad.abs().sum() > 0
    with pytest.raises(NotImplementedError):
        CameraMotionModule(ref, torch.zeros(1, 3, 4, 4), device="cpu", curve_type="spline")


def test_sh_colour_matches_reference_eval_sh():
    g = load("sh_golden.npz")
    dirs, sh = torch.tensor(g["dirs"]), torch.tensor(g["sh"])
    for deg in range(4):
        M = (deg + 1) ** 2
        mine = torch_naive.eval_sh_color(deg, sh[:, :M], dirs).numpy()
        assert np.abs(mine - g[f"deg{deg}"]).max() <= 2e-6, deg
    # the C++ oracle's computeColorFromSH: place Gaussians along the golden directions in front of the camera
    z = 4.0
    P = dirs.shape[0]
    d = g["dirs"].copy()
    d[:, 2] = np.abs(d[:, 2]) + 0.3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sc = synthetic.make_scene(P, 128, 128, K=1, seed=0, sh_degree=3)
    sc["means3D"] = (d * z).astype(np.float32)
    sc["sh"] = g["sh"]
    sc["campos"][0] = 0
    sc["viewmatrix"][0] = np.eye(4, dtype=np.float32)
    sc["projmatrix"][0] = sc["projection_matrix"]
    st = oracle_forward(sc, 0, sh_degree=3, render=False)
    vis = st["radii"] > 0
    assert vis.sum() >= 8
    ref = np.maximum(torch_naive.eval_sh_color(3, sh, torch.tensor(d.astype(np.float32))).numpy() + 0.5, 0)
    assert np.abs(st["rgb"][vis] - ref[vis]).max() <= 2e-6


def test_cov3d_matches_reference_build_scaling_rotation():
    g = load("cov3d_golden.npz")
    q = g["q"] / np.linalg.norm(g["q"], axis=1, keepdims=True)   # the reference normalises in Python
    P = q.shape[0]
    sc = synthetic.make_scene(P, 64, 64, K=1, seed=0)
    sc["means3D"][:] = [0, 0, 5]
    sc["scales"] = g["s"]
    sc["rotations"] = q.astype(np.float32)
    st = oracle_forward(sc, 0, render=False, scale_modifier=float(g["mod"]))
    rel = np.abs(st["cov3D"] - g["cov6"]).max() / np.abs(g["cov6"]).max()
    assert rel <= 2e-6
    R = torch_naive.quat_to_rotmat(torch.tensor(q.astype(np.float32))).numpy()
    assert np.abs(R - g["R"]).max() <= 1e-6


def test_projection_and_view_conventions():
    from deblurgs_amd import pose
    g = load("graphics_golden.npz")
    zn, zf, fx, fy = g["proj_args"]
    assert np.array_equal(pose.get_projection_matrix(zn, zf, fx, fy).numpy(), g["proj"])
    assert np.abs(synthetic.projection_matrix(zn, zf, fx, fy) - g["proj"]).max() <= 1e-7
    # a point transformed with the transposed (row-vector) matrices, as the kernels consume them
    Rt = g["w2v"]
    p = np.array([0.3, -0.2, 1.7, 1.0], np.float32)
    view_T = Rt.T
    assert np.allclose(p @ view_T, Rt @ p, atol=1e-6)
    assert abs(float(g["fov2focal"]) - 1920 / (2 * math.tan(fx / 2))) < 1e-9


def test_losses_match_reference_values_and_grads():
    from deblurgs_amd import losses
    g = load("loss_golden.npz")
    sub = torch.tensor(g["sub"], requires_grad=True)
    dep = torch.tensor(g["dep"], requires_grad=True)
    opa = torch.tensor(g["opa"], requires_grad=True)
    gt = torch.tensor(g["gt"])
    lam_t, lam_tv, lam_h = g["lam"]
    l1 = losses.l1_loss(sub.mean(0), gt)
    sm = losses.batchwise_smoothness_loss(sub)
    tv = losses.tv_loss(dep[:, None, :, :])
    hg = losses.hinge_l2(opa)
    total = l1 + lam_t * sm + lam_tv * tv + lam_h * hg
    total.backward()
    for mine, ref in [(l1, "l1"), (sm, "smooth"), (tv, "tv"), (hg, "hinge"), (total, "total")]:
        assert abs(float(mine) - float(g[ref])) <= 1e-7, ref
    assert np.abs(sub.grad.numpy() - g["g_sub"]).max() <= 1e-9
    assert np.abs(dep.grad.numpy() - g["g_dep"]).max() <= 1e-9
    assert np.abs(opa.grad.numpy() - g["g_opa"]).max() <= 1e-9
    assert float(losses.batchwise_smoothness_loss(sub[:1].detach())) == float(g["smooth_k1"][0])
    tot2, blur, a, b = losses.blur_loss_torch(torch.tensor(g["sub"]), gt, float(lam_t))
    assert abs(float(a) - float(g["l1"])) <= 1e-7 and abs(float(b) - float(g["smooth"])) <= 1e-7


def test_activations_match_reference():
    from deblurgs_amd import cloud, losses
    g = load("activation_golden.npz")
    x = torch.tensor(g["x"])
    assert np.array_equal(cloud.Clamp()(x).numpy(), g["clamp"])
    assert np.allclose(cloud.LowerBoundExponent(0.0)(x).numpy(), g["lbexp"], rtol=1e-7)
    img = torch.tensor(g["img"])
    tm = losses.ToneMapping("gamma")
    assert np.allclose(tm(img).numpy(), g["gamma"], rtol=1e-6)
    assert np.allclose(tm.inverse()(img).numpy(), g["inv_gamma"], rtol=1e-6)
    assert np.allclose(torch.nn.functional.normalize(x.reshape(10, 5)[:, :4]).numpy(), g["normalize"], rtol=1e-6)


def test_bezier_closed_form_and_numpy_twin():
    from deblurgs_amd import pose
    torch.manual_seed(0)
    b = pose.BezierModel(torch.randn(3, 3), 4, initial_noise=0.1)
    t = torch.tensor([0.0, 0.25, 1.0])
    out = b(t, 1)
    ctrl = b._control_points[1].detach().double().numpy()
    # control point 0 is reached at t = 1, the last one at t = 0 (scene/bezier.py:54-64)
    assert np.allclose(out[2].detach().numpy(), ctrl[0], atol=1e-6)
    assert np.allclose(out[0].detach().numpy(), ctrl[-1], atol=1e-6)
    assert np.allclose(out.detach().numpy(), synthetic.bezier_np(ctrl, t.numpy().astype(np.float64)), atol=1e-6)
    assert out.dtype == torch.float64     # the reference's binomial table is float64 (scene/bezier.py:48)


def test_pose_to_cameras_matches_reference_recipe():
    """scene/motion.py:258-294 written as a Python loop vs the batched product code."""
    from deblurgs_amd import pose
    torch.manual_seed(1)
    se3 = torch.randn(5, 6, dtype=torch.float64) * 0.1
    P = pose.get_projection_matrix(0.01, 100.0, 1.0, 0.7).transpose(0, 1)
    wv, fp, cc = pose.se3_to_view_proj(se3, P)
    c2w = pose.se3_exp_map(se3)
    for i in range(5):
        rot = c2w[i, :3, :3].transpose(-2, -1)
        trans = c2w[i, 3, :3]
        w = torch.eye(4)
        w[:3, :3] = rot
        w[3, :3] = -trans @ rot
        f = (w.unsqueeze(0).bmm(P.unsqueeze(0))).squeeze(0)
        assert torch.allclose(wv[i], w, atol=1e-7) and torch.allclose(fp[i], f, atol=1e-6)
        assert torch.allclose(cc[i], torch.inverse(w)[3][:3], atol=1e-6)


def test_oracle_backward_matches_float64_autograd():
    """Every analytic gradient of the oracle's backward (and the reference's partial dL_dviewmatrix) against
    autograd through the independent dense torch rasteriser in float64."""
    sc = synthetic.make_scene(400, 80, 64, K=2, seed=3)
    k = 1
    st = oracle_forward(sc, k)
    rng = np.random.default_rng(1)
    gC = rng.normal(size=(3, sc["H"], sc["W"])).astype(np.float32)
    gD = (rng.normal(size=(1, sc["H"], sc["W"])) * 0.1).astype(np.float32)
    gr = oracle.backward(st, gC, gD)
    dt = torch.float64
    T = lambda a: torch.tensor(a, dtype=dt)
    inp = {n: T(sc[n]).requires_grad_(True) for n in ["means3D", "opacities", "sh", "scales", "rotations"]}
    V = T(sc["viewmatrix"][k]).requires_grad_(True)
    F = T(sc["projmatrix"][k]).requires_grad_(True)
    c, d, r = torch_naive.rasterize(inp["means3D"], inp["opacities"], V, F, T(sc["campos"][k]), T(sc["bg"]), sc["W"],
                                    sc["H"], sc["tanfovx"], sc["tanfovy"], sh=inp["sh"], scales=inp["scales"],
                                    rotations=inp["rotations"], sh_degree=2)
    assert np.abs(c.detach().numpy() - st["color"]).max() <= 1e-5
    assert np.array_equal(r.numpy(), st["radii"])
    ((c * T(gC)).sum() + (d * T(gD)).sum()).backward()

    def rel(a, b):
        b = b.numpy()
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    assert rel(gr["dL_dmeans3D"], inp["means3D"].grad) <= 2e-4
    assert rel(gr["dL_dopacity"], inp["opacities"].grad) <= 2e-4
    assert rel(gr["dL_dsh"], inp["sh"].grad) <= 2e-4
    assert rel(gr["dL_dscales"], inp["scales"].grad) <= 2e-4
    assert rel(gr["dL_drotations"], inp["rotations"].grad) <= 2e-4
    assert rel(gr["dL_dviewmatrix"], V.grad) <= 2e-4
    # backward.cu:430-450: columns 0/1 of dL_dproj are the analytic gradient times 0.5*W / 0.5*H
    Fg, Pg = F.grad.numpy(), gr["dL_dprojmatrix"]
    assert np.allclose(Pg[:, 0], Fg[:, 0] * 0.5 * sc["W"], rtol=2e-4)
    assert np.allclose(Pg[:, 1], Fg[:, 1] * 0.5 * sc["H"], rtol=2e-4)
    assert np.all(Pg[:, 2] == 0) and np.all(Pg[:, 3] == Pg[0, 3])


def test_oracle_variants_run_and_are_consistent():
    sc = synthetic.make_scene(300, 64, 48, K=1, seed=5)
    a = oracle_forward(sc, 0)
    # precomputed colours equal to the SH colours give the same image
    b = oracle_forward(sc, 0, colors_precomp=a["rgb"].copy())
    assert np.array_equal(a["color"], b["color"])
    cov = a["cov3D"].copy()
    cov[a["depths"] == 0] = [1e-4, 0, 0, 1e-4, 0, 1e-4]
    c = oracle_forward(sc, 0, cov3D_precomp=cov)
    assert np.array_equal(a["color"], c["color"]) and np.array_equal(a["point_list"], c["point_list"])
    assert oracle.higher_msb(8160) == 13 and oracle.higher_msb(256) == 9 and oracle.higher_msb(1) == 1
