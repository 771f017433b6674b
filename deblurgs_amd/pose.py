"""Pose path of the blur-integration loop: Bezier curve in se(3) -> SE(3) -> rasteriser cameras.

Same names, argument meaning and error behaviour as the reference's
  scene/bezier.py:14-85                      BezierModel
  utils/pytorch3d_functions.py:373-540       se3_exp_map / se3_log_map / so3_exp_map / so3_log_map
  scene/motion.py:258-294                    _c2w_to_minicam  (batched here: no Python loop over K)
  scene/cameras.py:63-74                     MiniCam
  utils/graphics_utils.py:51-71              getProjectionMatrix
The exponential / logarithm maps are this package's own formulation (Rodrigues pieces shared between the maps); they run
at initialisation and on the CPU -- the training step evaluates the pose path on the device (csrc/pose.hip).  Both are
checked against tests/golden/pose_golden.npz (generated from the reference's own module) and against the function-by-
function restatement of the reference kept as test infrastructure in oracle/pose_oracle.py.
"""
import math

import torch
import torch.nn as nn


def _cross_matrices(w: torch.Tensor) -> torch.Tensor:
    """[N,3] -> [N,3,3] with  K(w) x = w cross x."""
    if w.ndim != 2 or w.shape[1] != 3:
        raise ValueError("Input vectors have to be 3-dimensional.")
    K = w.new_zeros(w.shape[0], 3, 3)
    K[:, 0, 1], K[:, 0, 2] = -w[:, 2], w[:, 1]
    K[:, 1, 0], K[:, 1, 2] = w[:, 2], -w[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -w[:, 1], w[:, 0]
    return K


def _rodrigues(w: torch.Tensor, eps: float):
    """The pieces of Rodrigues' formula for a batch of rotation vectors: the angle (its SQUARE clamped from below by eps,
    the reference's guard against the 0/0 at the identity, utils/pytorch3d_functions.py:236-238), K(w) and K(w)^2."""
    theta = (w * w).sum(dim=1).clamp(min=eps).sqrt()
    K = _cross_matrices(w)
    return theta, K, K @ K


def so3_exp_map(log_rot: torch.Tensor, eps: float = 0.0001) -> torch.Tensor:
    """Rotation vectors [N,3] -> rotation matrices [N,3,3]:  R = I + sin(t)/t K + (1 - cos t)/t^2 K^2."""
    if log_rot.ndim != 2 or log_rot.shape[1] != 3:
        raise ValueError("Input tensor shape has to be Nx3.")
    theta, K, K2 = _rodrigues(log_rot, eps)
    inv = 1.0 / theta
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)
    return eye + (inv * theta.sin())[:, None, None] * K + (inv * inv * (1.0 - theta.cos()))[:, None, None] * K2


def _left_jacobian(theta, K, K2):
    """V = I + (1 - cos t)/t^2 K + (t - sin t)/t^3 K^2: maps the translation logarithm to the translation."""
    eye = torch.eye(3, dtype=K.dtype, device=K.device)
    return (eye + ((1.0 - theta.cos()) / theta ** 2)[:, None, None] * K
            + ((theta - theta.sin()) / theta ** 3)[:, None, None] * K2)


def se3_exp_map(log_transform: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """[N,6] (translation logarithm | rotation vector) -> [N,4,4] rigid transforms in the row-vector convention
    [[R^T, 0], [t, 1]] the reference uses (utils/pytorch3d_functions.py:373-457; checked against tests/golden/
    pose_golden.npz and oracle/pose_oracle.py).  The training step itself evaluates this on the device
    (dgs_pose_forward, csrc/pose.hip); this is the initialisation / CPU implementation."""
    if log_transform.ndim != 2 or log_transform.shape[1] != 6:
        raise ValueError("Expected input to be of shape (N, 6).")
    u, w = log_transform[:, :3], log_transform[:, 3:]
    theta, K, K2 = _rodrigues(w, eps)
    inv = 1.0 / theta
    eye = torch.eye(3, dtype=w.dtype, device=w.device)
    R = eye + (inv * theta.sin())[:, None, None] * K + (inv * inv * (1.0 - theta.cos()))[:, None, None] * K2
    t = (_left_jacobian(theta, K, K2) @ u[:, :, None])[:, :, 0]
    out = log_transform.new_zeros(log_transform.shape[0], 4, 4)
    out[:, :3, :3] = R.transpose(1, 2)
    out[:, 3, :3] = t
    out[:, 3, 3] = 1.0
    return out


def _acos_guarded(x: torch.Tensor, bound: float) -> torch.Tensor:
    """acos on [-bound, bound], continued outside by its tangent line at +-bound: finite slope for traces that rounding
    pushed to (or past) 3 and -1 (the reference's acos_linear_extrapolation, utils/pytorch3d_functions.py:26-81)."""
    slope = -1.0 / math.sqrt(1.0 - bound * bound)
    hi = math.acos(bound) + (x - bound) * slope
    lo = math.acos(-bound) + (x + bound) * slope
    return torch.where(x >= bound, hi, torch.where(x <= -bound, lo, torch.acos(x.clamp(-bound, bound))))


def so3_log_map(R: torch.Tensor, eps: float = 0.0001, cos_bound: float = 1e-4) -> torch.Tensor:
    """Rotation matrices [N,3,3] -> rotation vectors [N,3]: the angle from the trace (guarded acos), then
    w = t / (2 sin t) * vee(R - R^T), with the series 1/2 + t^2/12 for that factor where |sin t| <= eps / 2
    (utils/pytorch3d_functions.py:121-176, 248-300)."""
    if R.ndim != 3 or R.shape[1:] != (3, 3):
        raise ValueError("Input has to be a batch of 3x3 Tensors.")
    trace = R.diagonal(dim1=1, dim2=2).sum(dim=1)
    if bool(((trace < -1.0 - eps) | (trace > 3.0 + eps)).any()):
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    c = 0.5 * (trace - 1.0)
    theta = _acos_guarded(c, 1.0 - cos_bound) if cos_bound > 0.0 else torch.acos(c)
    s = theta.sin()
    regular = s.abs() > 0.5 * eps
    factor = torch.where(regular, theta / (2.0 * torch.where(regular, s, torch.ones_like(s))),
                         0.5 + theta * theta / 12.0)
    A = factor[:, None, None] * (R - R.transpose(1, 2))
    if float((A + A.transpose(1, 2)).abs().max()) > 1e-5:
        raise ValueError("One of input matrices is not skew-symmetric.")
    return torch.stack((A[:, 2, 1], A[:, 0, 2], A[:, 1, 0]), dim=1)


def se3_log_map(transform: torch.Tensor, eps: float = 1e-4, cos_bound: float = 1e-4) -> torch.Tensor:
    """[N,4,4] rigid transforms (row-vector convention, last column (0,0,0,1)) -> [N,6] (translation logarithm | rotation
    vector): the inverse of se3_exp_map (utils/pytorch3d_functions.py:462-540).  Used once, at initialisation, on the
    dataset's camera poses (scene/motion.py:196-205)."""
    if transform.ndim != 3 or transform.shape[1:] != (4, 4):
        raise ValueError("Input tensor shape has to be (N, 4, 4).")
    if not torch.allclose(transform[:, :3, 3], torch.zeros_like(transform[:, :3, 3])):
        raise ValueError("All elements of `transform[:, :3, 3]` should be 0.")
    w = so3_log_map(transform[:, :3, :3].transpose(1, 2), eps=eps, cos_bound=cos_bound)
    theta, K, K2 = _rodrigues(w, eps)
    u = torch.linalg.solve(_left_jacobian(theta, K, K2), transform[:, 3, :3][:, :, None])[:, :, 0]
    return torch.cat((u, w), dim=1)


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
