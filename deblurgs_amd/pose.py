"""Pose path of the blur-integration loop: Bezier curve in se(3) -> SE(3) -> rasteriser cameras.

Mirrors (behaviour, names and argument meaning) of the reference's
  scene/bezier.py:14-85                      BezierModel
  utils/pytorch3d_functions.py:218-247       _so3_exp_map
  utils/pytorch3d_functions.py:337-372       hat
  utils/pytorch3d_functions.py:373-457       se3_exp_map
  utils/pytorch3d_functions.py:546-573       _se3_V_matrix
  scene/motion.py:258-294                    _c2w_to_minicam  (batched here: no Python loop over K)
  scene/cameras.py:63-74                     MiniCam
  utils/graphics_utils.py:51-71              getProjectionMatrix
The se3/so3 maps are pinned by tests/golden/pose_golden.npz (generated from the reference's own module).
"""
import math

import torch
import torch.nn as nn


def hat(v: torch.Tensor) -> torch.Tensor:
    """Skew-symmetric matrices of a batch of 3-vectors (pytorch3d_functions.py:337-372)."""
    N, dim = v.shape
    if dim != 3:
        raise ValueError("Input vectors have to be 3-dimensional.")
    x, y, z = v.unbind(1)
    zero = torch.zeros_like(x)
    return torch.stack((zero, -z, y, z, zero, -x, -y, x, zero), dim=1).reshape(N, 3, 3)


def _so3_exp_map(log_rot: torch.Tensor, eps: float = 0.0001):
    _, dim = log_rot.shape
    if dim != 3:
        raise ValueError("Input tensor shape has to be Nx3.")
    nrms = (log_rot * log_rot).sum(1)
    rot_angles = torch.clamp(nrms, eps).sqrt()
    rot_angles_inv = 1.0 / rot_angles
    fac1 = rot_angles_inv * rot_angles.sin()
    fac2 = rot_angles_inv * rot_angles_inv * (1.0 - rot_angles.cos())
    skews = hat(log_rot)
    skews_square = torch.bmm(skews, skews)
    R = (fac1[:, None, None] * skews + fac2[:, None, None] * skews_square
         + torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None])
    return R, rot_angles, skews, skews_square


def so3_exp_map(log_rot: torch.Tensor, eps: float = 0.0001) -> torch.Tensor:
    return _so3_exp_map(log_rot, eps=eps)[0]


def _se3_V_matrix(log_rotation, log_rotation_hat, log_rotation_hat_square, rotation_angles, eps: float = 1e-4):
    return (torch.eye(3, dtype=log_rotation.dtype, device=log_rotation.device)[None]
            + log_rotation_hat * ((1 - torch.cos(rotation_angles)) / (rotation_angles ** 2))[:, None, None]
            + log_rotation_hat_square
            * ((rotation_angles - torch.sin(rotation_angles)) / (rotation_angles ** 3))[:, None, None])


def se3_exp_map(log_transform: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """[N,6] (log_translation | log_rotation) -> [N,4,4] SE(3) in the row-vector convention [[R,0],[T,1]]."""
    if log_transform.ndim != 2 or log_transform.shape[1] != 6:
        raise ValueError("Expected input to be of shape (N, 6).")
    N, _ = log_transform.shape
    log_translation = log_transform[..., :3]
    log_rotation = log_transform[..., 3:]
    R, rotation_angles, log_rotation_hat, log_rotation_hat_square = _so3_exp_map(log_rotation, eps=eps)
    V = _se3_V_matrix(log_rotation, log_rotation_hat, log_rotation_hat_square, rotation_angles, eps=eps)
    T = torch.bmm(V, log_translation[:, :, None])[:, :, 0]
    transform = torch.zeros(N, 4, 4, dtype=log_transform.dtype, device=log_transform.device)
    transform[:, :3, :3] = R
    transform[:, :3, 3] = T
    transform[:, 3, 3] = 1.0
    return transform.permute(0, 2, 1)


def _acos_linear_extrapolation(x, bound):
    """acos(x) inside (-bound, bound), its first-order Taylor line outside (pytorch3d_functions.py:26-81)."""
    def line(x0):
        return (x - x0) * (-1.0 / math.sqrt(1.0 - x0 * x0)) + math.acos(x0)
    inner = torch.acos(x.clamp(-bound, bound))
    return torch.where(x >= bound, line(bound), torch.where(x <= -bound, line(-bound), inner))


def so3_log_map(R: torch.Tensor, eps: float = 0.0001, cos_bound: float = 1e-4) -> torch.Tensor:
    """[N,3,3] rotation matrices -> [N,3] logarithms (pytorch3d_functions.py:248-300 with so3_rotation_angle
    :121-176 and hat_inv :303-336): angle from the trace through the linearly extrapolated acos, factor
    phi / (2 sin phi) with its second-order Taylor value where |sin phi| <= eps / 2."""
    if R.ndim != 3 or R.shape[1:] != (3, 3):
        raise ValueError("Input has to be a batch of 3x3 Tensors.")
    rot_trace = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    if ((rot_trace < -1.0 - eps) + (rot_trace > 3.0 + eps)).any():
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    phi_cos = (rot_trace - 1.0) * 0.5
    phi = _acos_linear_extrapolation(phi_cos, 1.0 - cos_bound) if cos_bound > 0.0 else torch.acos(phi_cos)
    phi_sin = torch.sin(phi)
    ok = phi_sin.abs() > (0.5 * eps)
    safe = torch.where(ok, phi_sin, torch.ones_like(phi_sin))
    phi_factor = torch.where(ok, phi / (2.0 * safe), 0.5 + (phi ** 2) * (1.0 / 12))
    h = phi_factor[:, None, None] * (R - R.permute(0, 2, 1))
    if float(torch.abs(h + h.permute(0, 2, 1)).max()) > 1e-5:
        raise ValueError("One of input matrices is not skew-symmetric.")
    return torch.stack((h[:, 2, 1], h[:, 0, 2], h[:, 1, 0]), dim=1)


def se3_log_map(transform: torch.Tensor, eps: float = 1e-4, cos_bound: float = 1e-4) -> torch.Tensor:
    """[N,4,4] SE(3) matrices in the row-vector convention [[R,0],[T,1]] -> [N,6] (log_translation | log_rotation)
    (pytorch3d_functions.py:462-540): so3_log_map of the transposed upper-left block, translation = V^-1 T.
    Used once, at initialisation, on the dataset's camera poses (scene/motion.py:196-205)."""
    if transform.ndim != 3 or transform.shape[1:] != (4, 4):
        raise ValueError("Input tensor shape has to be (N, 4, 4).")
    if not torch.allclose(transform[:, :3, 3], torch.zeros_like(transform[:, :3, 3])):
        raise ValueError("All elements of `transform[:, :3, 3]` should be 0.")
    log_rotation = so3_log_map(transform[:, :3, :3].permute(0, 2, 1), eps=eps, cos_bound=cos_bound)
    nrms = (log_rotation ** 2).sum(-1)
    rotation_angles = torch.clamp(nrms, eps).sqrt()
    skew = hat(log_rotation)
    V = _se3_V_matrix(log_rotation, skew, torch.bmm(skew, skew), rotation_angles, eps=eps)
    log_translation = torch.linalg.solve(V, transform[:, 3, :3][:, :, None])[:, :, 0]
    return torch.cat((log_translation, log_rotation), dim=1)


# ---- unit quaternions <-> rotation matrices for curve_type="quarternion_cartesian" (scene/motion.py:191-194,242-246).
# The reference calls the third-party `roma` (unpinned in environment.yml:24, absent from /root/reference and from this
# image): roma.rotmat_to_unitquat / roma.unitquat_to_rotmat, XYZW convention.  roma documents rotmat_to_unitquat as
# adapted from SciPy's Rotation.from_matrix(...).as_quat(); that published algorithm is restated here and
# tests/test_oracle_golden.py pins it against scipy (available in this image) -- "parity unpinned" against roma itself.
def rotmat_to_unitquat(R: torch.Tensor) -> torch.Tensor:
    """[N,3,3] -> [N,4] unit quaternions (x, y, z, w): pick the largest of (m00, m11, m22, trace) as pivot so that
    no component is formed from a cancelling difference, then normalise.  No sign canonicalisation (as roma / scipy)."""
    if R.ndim != 3 or R.shape[1:] != (3, 3):
        raise ValueError("Input has to be a batch of 3x3 Tensors.")
    m = R
    trace = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    decision = torch.stack([m[:, 0, 0], m[:, 1, 1], m[:, 2, 2], trace], dim=1)
    choice = decision.argmax(dim=1)
    quat = torch.empty((R.shape[0], 4), dtype=R.dtype, device=R.device)
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        q = torch.empty_like(quat)
        q[:, i] = 1 - trace + 2 * m[:, i, i]
        q[:, j] = m[:, j, i] + m[:, i, j]
        q[:, k] = m[:, k, i] + m[:, i, k]
        q[:, 3] = m[:, k, j] - m[:, j, k]
        quat = torch.where((choice == i)[:, None], q, quat)
    q = torch.stack([m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1], 1 + trace], dim=1)
    quat = torch.where((choice == 3)[:, None], q, quat)
    return quat / quat.norm(dim=1, keepdim=True)


def unitquat_to_rotmat(quat: torch.Tensor) -> torch.Tensor:
    """[N,4] unit quaternions (x, y, z, w) -> [N,3,3]."""
    x, y, z, w = quat.unbind(-1)
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    return torch.stack([x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw),
                        2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw),
                        2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2], dim=-1).reshape(quat.shape[:-1] + (3, 3))


def _binom(n: int, k: int) -> float:
    return float(math.comb(n, k))


class BezierModel(nn.Module):
    """Bernstein-basis Bezier curves, one per training image (scene/bezier.py).

    Note the reference's parametrisation: control point 0 is reached at t = 1
    (coeff_k = binom(C,k) * t^(C-k) * (1-t)^k, scene/bezier.py:54-64)."""

    def __init__(self, initial_points, curve_order, initial_noise=0.001, device=None):
        super().__init__()
        self.curve_order = curve_order
        initial_points = initial_points.float()
        if device is not None:
            initial_points = initial_points.to(device)
        initial_points = initial_points[:, None, :].repeat(1, curve_order + 1, 1)
        initial_points = initial_points + torch.randn_like(initial_points) * initial_noise
        self._control_points = nn.Parameter(initial_points.clone().contiguous().requires_grad_(True))
        # The reference builds this with scipy.special.binom as a float64 tensor (scene/bezier.py:48), which
        # promotes coeff to float64; math.comb gives the same integers.
        self.register_buffer(
            "_bezier_binom_coeff",
            torch.tensor([_binom(curve_order, k) for k in range(curve_order + 1)], dtype=torch.float64,
                         device=self._control_points.device), persistent=False)

    @property
    def device(self):
        return self._control_points.device

    def _get_bezier_coeff(self, t):
        C = self.curve_order
        coeff = ((t[:, None] ** torch.arange(C, -1, -1, device=self.device))
                 * ((1 - t)[:, None] ** torch.arange(0, C + 1, device=self.device))
                 * self._bezier_binom_coeff.to(self.device))
        return coeff

    def forward(self, t: torch.Tensor, idx):
        if isinstance(idx, int):
            idx = torch.tensor([idx], device=self.device)
        return (self._get_bezier_coeff(t)[:, :, None] * self._control_points[idx]).sum(dim=1)

    def __len__(self):
        return self._control_points.shape[0]


def get_projection_matrix(znear, zfar, fovX, fovY):
    """utils/graphics_utils.py:51-71 (returned un-transposed, like the reference)."""
    tanHalfFovY = math.tan((fovY / 2))
    tanHalfFovX = math.tan((fovX / 2))
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class MiniCam:
    """scene/cameras.py:63-74 -- the four tensors render() reads plus the intrinsics."""

    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform,
                 camera_center=None):
        self.image_width = width
        self.image_height = height
        self.FoVy = fovy
        self.FoVx = fovx
        self.znear = znear
        self.zfar = zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        if camera_center is None:
            view_inv = torch.inverse(self.world_view_transform)
            camera_center = view_inv[3][:3]
        self.camera_center = camera_center


class _FusedTrajectory(torch.autograd.Function):
    """(ctrl_trans [C+1,3], ctrl_rot [C+1,3], nu [K], proj_T [4,4]) -> (world_view, full_proj, camera_center) through
    the single-kernel pose path of libdgs_hip.so (csrc/pose.hip); same math as BezierModel + se3_exp_map +
    c2w_to_view_proj, which stay as the torch reference implementation (and the CPU path of the tests).
    ctrl_rot [C+1,4] selects the quaternion + cartesian curves of curve_type="quarternion_cartesian"."""

    @staticmethod
    def forward(ctx, ctrl_trans, ctrl_rot, nu, proj_T):
        import ctypes
        from . import _lib
        L = _lib.lib()
        dev = ctrl_trans.device
        ct = ctrl_trans.detach().float().contiguous()
        cr = ctrl_rot.detach().float().contiguous()
        t = nu.detach().float().contiguous()
        pj = proj_T.detach().float().contiguous()
        C, K = ct.shape[0] - 1, t.shape[0]
        quat = int(cr.shape[-1] == 4)
        view = torch.empty((K, 4, 4), dtype=torch.float32, device=dev)
        full = torch.empty((K, 4, 4), dtype=torch.float32, device=dev)
        cam = torch.empty((K, 3), dtype=torch.float32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.dgs_pose_forward(ct.data_ptr(), cr.data_ptr(), t.data_ptr(), pj.data_ptr(), C, K, quat,
                                      view.data_ptr(), full.data_ptr(), cam.data_ptr(), st), "dgs_pose_forward")
        ctx.save_for_backward(ct, cr, t, pj)
        ctx.mark_non_differentiable(cam)
        return view, full, cam

    @staticmethod
    def backward(ctx, g_view, g_full, _g_cam):
        import ctypes
        from . import _lib
        L = _lib.lib()
        ct, cr, t, pj = ctx.saved_tensors
        dev = ct.device
        C, K = ct.shape[0] - 1, t.shape[0]
        gv = torch.zeros((K, 4, 4), dtype=torch.float32, device=dev) if g_view is None else g_view.float().contiguous()
        gf = torch.zeros((K, 4, 4), dtype=torch.float32, device=dev) if g_full is None else g_full.float().contiguous()
        d_ct, d_cr = torch.empty_like(ct), torch.empty_like(cr)
        d_nu = torch.empty_like(t)
        scratch = torch.empty(L.dgs_pose_scratch_bytes(K), dtype=torch.uint8, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.dgs_pose_backward(ct.data_ptr(), cr.data_ptr(), t.data_ptr(), pj.data_ptr(), C, K,
                                       int(cr.shape[-1] == 4), gv.data_ptr(),
                                       gf.data_ptr(), scratch.data_ptr(), d_ct.data_ptr(), d_cr.data_ptr(),
                                       d_nu.data_ptr(), st), "dgs_pose_backward")
        return d_ct, d_cr, d_nu, None


def fused_trajectory(ctrl_trans, ctrl_rot, nu, projection_matrix):
    """Single-kernel pose path (device tensors only)."""
    return _FusedTrajectory.apply(ctrl_trans, ctrl_rot, nu, projection_matrix)


def c2w_to_view_proj(rots, transes, projection_matrix):
    """Batched core of scene/motion.py:258-294: c2w rotations [K,3,3] + translations [K,3] ->
    (world_view [K,4,4], full_proj [K,4,4], camera_center [K,3]), differentiable w.r.t. rots/transes."""
    K = rots.shape[0]
    # The reference evaluates the curve and the exponential map in float64 (the binomial table is float64,
    # scene/bezier.py:48) and rounds to float32 when writing into torch.eye(4) (scene/motion.py:277-279).
    world_view = torch.eye(4, device=rots.device, dtype=torch.float32)[None].repeat(K, 1, 1)
    world_view[:, :3, :3] = rots.to(torch.float32)
    world_view[:, 3, :3] = (-torch.bmm(transes[:, None, :], rots)[:, 0, :]).to(torch.float32)
    full_proj = torch.matmul(world_view, projection_matrix.to(world_view)[None])
    campos = torch.linalg.inv(world_view)[:, 3, :3]
    return world_view, full_proj, campos


def se3_to_view_proj(se3, projection_matrix):
    """scene/motion.py:248-252 followed by :258-294, for all K subframes at once."""
    c2w = se3_exp_map(se3)
    rots = c2w[:, :3, :3].transpose(-2, -1)
    transes = c2w[:, 3, :3]
    return c2w_to_view_proj(rots, transes, projection_matrix)
