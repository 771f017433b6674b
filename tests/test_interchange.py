"""CPU tests of the interchange formats: Gaussian PLY, chkpnt tuple, cm.pth."""
import numpy as np
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
