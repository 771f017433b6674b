"""CPU tests of the interchange formats: Gaussian PLY, chkpnt tuple, cm.pth."""
import numpy as np
import pytest
import torch

from helpers import synthetic


def _cloud(P=50, deg=2, seed=0):
    from deblurgs_amd.cloud import GaussianCloud
    sc = synthetic.make_scene(P, 32, 32, K=1, seed=seed, sh_degree=deg)
    return GaussianCloud.from_scene(sc, "cpu")


def test_ply_attribute_order_matches_reference():
    from deblurgs_amd.interchange import ply_attributes
    # scene/gaussian_model.py:206-224 for sh_degree=2: 3 dc + 24 rest
    exp = (['x', 'y', 'z', 'nx', 'ny', 'nz', 'f_dc_0', 'f_dc_1', 'f_dc_2'] + [f'f_rest_{i}' for i in range(24)]
           + ['opacity', 'scale_0', 'scale_1', 'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3'])
    assert ply_attributes(24) == exp


def test_ply_round_trip_and_layout(tmp_path):
    from deblurgs_amd import interchange
    cloud = _cloud()
    with torch.no_grad():
        cloud._opacity.clamp_(0.02, 0.98)       # the stored logit is finite only inside (0, 1)
    path = str(tmp_path / "point_cloud.ply")
    interchange.save_ply(cloud, path)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n") + 11].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\n")
    assert len(raw) - len(head) == 50 * (6 + 3 + 24 + 1 + 3 + 4) * 4
    v = interchange._read_ply(path)
    # opacity is stored as a logit, scale as a log, SH channel-major
    assert np.allclose(1 / (1 + np.exp(-v["opacity"])), cloud.get_opacity.detach().numpy().reshape(-1), atol=1e-6)
    assert np.allclose(v["scale_1"], cloud._scaling.detach().numpy()[:, 1], atol=1e-6)
    assert np.allclose(v["f_rest_8"], cloud._features_rest.detach().numpy()[:, 0, 1])   # channel 1, coeff 0
    back = interchange.load_ply(path, sh_degree=2, device="cpu")
    for a, b in zip(cloud.hot_parameters(), back.hot_parameters()):
        assert torch.allclose(a, b, atol=2e-6), (a - b).abs().max()
    assert torch.allclose(back.get_features, cloud.get_features, atol=1e-6)


def test_checkpoint_and_camera_motion_round_trip(tmp_path):
    from deblurgs_amd import interchange
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    cloud = _cloud(30, 1)
    opt = torch.optim.Adam(cloud.hot_parameters(), lr=1e-3)
    capture = (1, cloud._xyz, cloud._features_dc, cloud._features_rest, cloud._scaling, cloud._rotation,
               cloud._opacity, torch.zeros(30), torch.zeros(30, 1), torch.zeros(30, 1), opt.state_dict(), 1.0)
    p = str(tmp_path / "chkpnt7000.pth")
    torch.save((capture, 7000), p)
    back, it, extras = interchange.load_checkpoint(p, device="cpu")
    assert it == 7000 and back.max_sh_degree == 1 and back.active_sh_degree == 1
    assert torch.equal(back._xyz, cloud._xyz) and "optimizer" in extras
    ref = RefCamera(32, 32, 1.0, 1.0, device="cpu")
    m = CameraMotionModule(ref, torch.rand(3, 3, 32, 32), curve_order=4, num_subframes=6, device="cpu")
    cm = str(tmp_path / "cm.pth")
    interchange.save_camera_motion(m, cm)
    m2 = CameraMotionModule(ref, torch.rand(3, 3, 32, 32), curve_order=4, num_subframes=6, device="cpu")
    interchange.load_camera_motion(m2, str(tmp_path))
    assert torch.equal(m2._rot._control_points, m._rot._control_points)
    assert torch.equal(m2._trans._control_points, m._trans._control_points) and torch.equal(m2._nu, m._nu)
    a = m.get_trajectory_matrices(2, fused=False)
    b = m2.get_trajectory_matrices(2, fused=False)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_capture_restore_round_trip(tmp_path):
    """GaussianCloud.capture() is the reference's checkpoint tuple; restore() rebuilds the cloud, its optimiser
    groups and the optimiser state from it (scene/gaussian_model.py:80-112)."""
    import types
    from deblurgs_amd import interchange
    cloud = _cloud(40, 2)
    targs = types.SimpleNamespace(iterations=1000, position_lr_init=1.6e-4, position_lr_final=1.6e-6, feature_lr=2.5e-3,
                                  opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3, percent_dense=0.01)
    cloud.training_setup(targs, spatial_lr_scale=2.0, fused=False)      # torch Adam: this test runs on the CPU
    for p in cloud.hot_parameters():
        p.grad = torch.randn_like(p) * 1e-2
    cloud.optimizer.step()
    cloud.xyz_gradient_accum += 0.5
    path = str(tmp_path / "chkpnt30.pth")
    interchange.save_checkpoint(cloud, 30, path)
    tup, it = torch.load(path, weights_only=False)
    assert it == 30 and len(tup) == 12 and tup[0] == cloud.active_sh_degree and tup[11] == 2.0
    other = _cloud(40, 2, seed=5)
    other.restore(tup, targs, fused=False)
    for a, b in zip(cloud.hot_parameters(), other.hot_parameters()):
        assert torch.equal(a, b)
        sa, sb = cloud.optimizer.state[a], other.optimizer.state[b]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and float(sa["step"]) == float(sb["step"])
    assert torch.equal(other.xyz_gradient_accum, cloud.xyz_gradient_accum) and other.spatial_lr_scale == 2.0
    back, it2, extras = interchange.load_checkpoint(path, device="cpu")
    assert it2 == 30 and torch.equal(back._xyz, cloud._xyz) and extras["spatial_lr_scale"] == 2.0


def _targs():
    import types
    return types.SimpleNamespace(iterations=150_000, position_lr_init=1.6e-4, position_lr_final=1.6e-6, feature_lr=2.5e-3,
                                 opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3, percent_dense=0.01)


def test_reference_written_checkpoint_loads_and_continues():
    """tests/golden/chkpnt_ref.pth is the file train.py:214-216 writes, produced by the REFERENCE's GaussianModel
    (capture() after three torch.optim.Adam steps; tests/golden/make_golden_r2.py).  load_checkpoint reads it,
    restore() rebuilds parameters + optimiser from it, and the next optimiser step on the recorded gradients lands on
    the reference's own next parameters and moments (chkpnt_ref_next.npz)."""
    import os
    import numpy as np
    from deblurgs_amd import interchange
    G = os.path.join(os.path.dirname(__file__), "golden")
    path = os.path.join(G, "chkpnt_ref.pth")
    nxt = np.load(os.path.join(G, "chkpnt_ref_next.npz"))
    cloud, it, extras = interchange.load_checkpoint(path, device="cpu")
    assert it == 103 and cloud.active_sh_degree == 1 and cloud.max_sh_degree == 2 and cloud._xyz.shape == (48, 3)
    assert extras["spatial_lr_scale"] == 2.5 and extras["max_radii2D"].shape == (48,)
    assert [g["name"] for g in extras["optimizer"]["param_groups"]] == ["xyz", "f_dc", "f_rest", "opacity", "scaling",
                                                                       "rotation"]
    tup, _ = torch.load(path, weights_only=False)
    other = _cloud(5, 2, seed=9)                      # a different cloud: everything must come from the tuple
    other.restore(tup, _targs(), fused=False)
    assert other._xyz.shape == (48, 3) and other.spatial_lr_scale == 2.5 and other.active_sh_degree == 1
    assert torch.equal(other.xyz_gradient_accum, tup[8]) and torch.equal(other.denom, tup[9])
    named = other._named()
    for n, p in named.items():
        p.grad = torch.from_numpy(nxt["grad_" + n])
    other.update_learning_rate(104)
    assert abs(other.optimizer.param_groups[0]["lr"] - float(nxt["lr_xyz"])) <= 1e-12
    other.optimizer.step()
    for n, p in named.items():
        assert np.abs(p.detach().numpy() - nxt["param_" + n]).max() <= 1e-7, n
        st = other.optimizer.state[p]
        assert np.abs(st["exp_avg"].numpy() - nxt["m_" + n]).max() <= 1e-9, n
        assert np.abs(st["exp_avg_sq"].numpy() - nxt["v_" + n]).max() <= 1e-12, n
        assert float(st["step"]) == float(nxt["step"])


def test_restore_with_trajectory_groups(tmp_path):
    """A checkpoint written during real training holds nine optimiser groups (six per-Gaussian + curve_rot / curve_trans
    / curve_alignment, scene/motion.py:63-76).  restore() attaches the motion module's groups before loading when the
    module is given, and loads only the per-Gaussian groups otherwise (the reference's own restore raises here)."""
    from deblurgs_amd import interchange
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    torch.manual_seed(3)
    ref = RefCamera(32, 32, 1.0, 1.0, device="cpu")
    cloud = _cloud(20, 2)
    cloud.training_setup(_targs(), spatial_lr_scale=1.0, fused=False)
    m = CameraMotionModule(ref, torch.rand(3, 3, 32, 32), curve_order=3, num_subframes=5, device="cpu")
    m.add_training_setup(cloud, {"curve_rot": 1e-3, "curve_trans": 1e-2, "curve_alignment": 1e-4})
    assert len(cloud.optimizer.param_groups) == 9
    for grp in cloud.optimizer.param_groups:
        for p in grp["params"]:
            p.grad = torch.randn_like(p) * 1e-2
    cloud.optimizer.step()
    path = str(tmp_path / "chkpnt9.pth")
    interchange.save_checkpoint(cloud, 9, path)
    tup, _ = torch.load(path, weights_only=False)
    assert len(tup[10]["param_groups"]) == 9
    # (a) with the motion module: all nine groups and their moments come back
    a = _cloud(20, 2, seed=4)
    m2 = CameraMotionModule(ref, torch.rand(3, 3, 32, 32), curve_order=3, num_subframes=5, device="cpu")
    a.restore(tup, _targs(), fused=False, cam_motion_module=m2)
    assert [g["name"] for g in a.optimizer.param_groups][6:] == ["curve_rot", "curve_trans", "curve_alignment"]
    assert a.optimizer.param_groups[7]["lr"] == 1e-2
    for p_old, p_new in zip(m.parameters(), m2.parameters()):
        assert torch.equal(cloud.optimizer.state[p_old]["exp_avg"], a.optimizer.state[p_new]["exp_avg"])
    # (b) without it: six groups, per-Gaussian moments intact, nothing raised
    b = _cloud(20, 2, seed=5)
    b.restore(tup, _targs(), fused=False)
    assert len(b.optimizer.param_groups) == 6
    assert torch.equal(b.optimizer.state[b._scaling]["exp_avg_sq"], cloud.optimizer.state[cloud._scaling]["exp_avg_sq"])
    assert len(tup[10]["param_groups"]) == 9            # the caller's tuple was not edited


def test_fused_adam_state_dict_loads_into_torch_adam():
    """FusedAdam's param_groups carry torch.optim.Adam's keys, so the reference's restore (torch.optim.Adam
    .load_state_dict + step) accepts a checkpoint written here; non-default values are rejected, not ignored."""
    from deblurgs_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(8))
    fo = FusedAdam([{"params": [p], "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15)
    sd = fo.state_dict()
    q = torch.nn.Parameter(torch.zeros(8))
    to = torch.optim.Adam([{"params": [q], "lr": 0.5, "name": "xyz"}], lr=0.0, eps=1e-15)
    to.load_state_dict(sd)
    q.grad = torch.ones(8)
    to.step()                                            # KeyError 'weight_decay' before the defaults were added
    assert torch.allclose(q.detach(), torch.full((8,), -0.1)) and to.param_groups[0]["lr"] == 0.1
    fo.param_groups[0]["amsgrad"] = True
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="amsgrad"):
        fo.step()
