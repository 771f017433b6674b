"""Blur-integration loop: per-image learnable SE(3) Bezier trajectory and the K-subframe query.

Mirrors the hot part of the reference's scene/motion.py (CameraMotionModule.query :78-160, get_trajectory
:162-178, _sample_nu_from_alignment :209-219, _sample_c2w_from_nu :221-256, _c2w_to_minicam :258-294) for both
curve types ("se3", the default, and "quarternion_cartesian").  Differences, all on purpose:
  * the K subframes are rasterised by ONE fused launch chain (gaussian_renderer.render_subframes) instead of a
    Python loop of K render() calls, and the K cameras are built with batched tensor ops;
  * dataset loading, PLY/COLMAP I/O, optimiser wiring and cm.pth checkpoints are out of scope (SURVEY 2, rows
    12-21): the module is constructed from initial c2w poses and ground-truth images held in memory;
  * curve_type="quarternion_cartesian" calls the third-party `roma` in the reference (unpinned, absent here); its two
    conversions are restated in pose.py (rotmat_to_unitquat / unitquat_to_rotmat) and pinned against scipy; on the
    device both curve types go through the fused pose kernel (csrc/pose.hip), the torch ops are its test reference.
"""
import torch
import torch.nn as nn

from . import gaussian_renderer
from .pose import (BezierModel, MiniCam, get_projection_matrix, se3_exp_map, se3_log_map, c2w_to_view_proj,
                   fused_trajectory, rotmat_to_unitquat, unitquat_to_rotmat)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class RefCamera:
    """The attributes of the reference Camera that _c2w_to_minicam copies (scene/motion.py:283-292)."""

    def __init__(self, width, height, fovx, fovy, znear=0.01, zfar=100.0, device="cuda"):
        self.image_width = width
        self.image_height = height
        self.FoVx = fovx
        self.FoVy = fovy
        self.znear = znear
        self.zfar = zfar
        self.projection_matrix = get_projection_matrix(znear=znear, zfar=zfar, fovX=fovx, fovY=fovy).transpose(0, 1).to(device)


class CameraMotionModule:
    def __init__(self, ref_cam: RefCamera, gt_images, curve_order=9, num_subframes=21, init_se3=None,
                 curve_random_sample=False, device="cuda", curve_type="se3", init_c2w=None):
        """gt_images: [n,3,H,W] observed blurry images.  Initial poses, one per image, either as
        init_c2w = (rotations [n,3,3], translations [n,3]) -- the c2w rotation and camera position the reference takes
        from its CameraInfo list (scene/motion.py:39-50: cam_info.R and -T @ R^T) -- or as init_se3 [n,6]
        (trans_log | rot_log) logarithms; identity poses when neither is given."""
        if curve_type not in ("se3", "quarternion_cartesian"):
            raise NotImplementedError(curve_type)            # scene/motion.py:206-207
        self.curve_order = curve_order
        self.n_subframes = num_subframes
        self.curve_type = curve_type
        self.curve_random_sample = curve_random_sample
        self.gaussians = None
        self.ref_cam = ref_cam
        self.gt_images = gt_images
        n = gt_images.shape[0]
        if init_c2w is None:
            if init_se3 is None:
                init_se3 = torch.zeros(n, 6)
            c2w = se3_exp_map(init_se3.double())
            init_c2w = (c2w[:, :3, :3].transpose(-2, -1), c2w[:, 3, :3])
        elif init_se3 is not None:
            raise ValueError("give init_se3 or init_c2w, not both")
        self._set_initial_parameters(init_c2w[0], init_c2w[1], device, init_se3)
        f = num_subframes
        if f > 2:
            nu0 = torch.linspace(1 / (f - 1), 1.0 - (1 / (f - 1)), f - 2)[None, :].repeat(n, 1).to(device)
            self._nu = nn.Parameter(inverse_sigmoid(nu0).contiguous().requires_grad_(True))
        else:
            self._nu = nn.Parameter(torch.zeros(n, 0, device=device))

    def _set_initial_parameters(self, rotations, translations, device, init_se3=None):
        """scene/motion.py:180-207.  rotations: c2w rotation [n,3,3]; translations: camera position [n,3]."""
        n = rotations.shape[0]
        if self.curve_type == "quarternion_cartesian":
            self._rot = BezierModel(rotmat_to_unitquat(rotations), self.curve_order, device=device)
            self._trans = BezierModel(translations, self.curve_order, initial_noise=0.01, device=device)
            return
        if init_se3 is not None:
            params = init_se3            # already logarithms: skip the exp -> log round trip
        else:
            c2w = torch.zeros(n, 4, 4, dtype=rotations.dtype, device=rotations.device)
            c2w[:, :3, :3] = rotations.transpose(-2, -1)     # row-vector (torch3d) convention
            c2w[:, 3, :3] = translations
            c2w[:, 3, 3] = 1.0
            params = se3_log_map(c2w)
        self._rot = BezierModel(params[:, 3:], self.curve_order, device=device)
        self._trans = BezierModel(params[:, :3], self.curve_order, device=device)

    def link_gaussian(self, gaussians):
        self.gaussians = gaussians

    def add_training_setup(self, gaussians, lr_dict):
        """scene/motion.py:63-76: the curve parameters join the Gaussians' optimiser as the groups curve_rot,
        curve_trans and curve_alignment (FusedAdam updates them in the same launches as the per-Gaussian groups)."""
        opt = gaussians.optimizer
        for group in opt.param_groups:
            if "curve_" in group['name'] and group['params'][0] in opt.state:
                del opt.state[group['params'][0]]
        opt.param_groups = [e for e in opt.param_groups if 'curve_' not in e['name']]
        opt.add_param_group({'params': list(self._rot.parameters()), 'lr': lr_dict['curve_rot'], 'name': 'curve_rot'})
        opt.add_param_group({'params': list(self._trans.parameters()), 'lr': lr_dict['curve_trans'],
                             'name': 'curve_trans'})
        opt.add_param_group({'params': [self._nu], 'lr': lr_dict['curve_alignment'], 'name': 'curve_alignment'})

    def is_optimizing(self):
        return bool(self._nu.requires_grad)

    def alternate_optimization(self):
        """scene/motion.py:312-320."""
        new_state = not self.is_optimizing()
        for optimizable in [self._rot, self._trans, self._nu]:
            optimizable.requires_grad_(new_state)

    def parameters(self):
        return list(self._rot.parameters()) + list(self._trans.parameters()) + [self._nu]

    def __len__(self):
        return len(self._rot)

    @property
    def device(self):
        return self._rot.device

    # ---- scene/motion.py:209-219
    def _sample_nu_from_alignment(self, idx, uniform=None):
        """uniform: the U(0,1) jitter samples to use when curve_random_sample is on (default: drawn here); ranks that
        split ONE view's subframes must all use the same draw."""
        device = self._nu.device
        nu_mid = torch.sigmoid(self._nu[idx])
        if self.curve_random_sample:
            u = torch.rand_like(nu_mid) if uniform is None else uniform.to(nu_mid).reshape(nu_mid.shape)
            nu_mid = nu_mid + u / self.n_subframes - (1 / (2 * self.n_subframes))
        return torch.cat([torch.zeros(1, device=device), nu_mid, torch.ones(1, device=device)]).clamp(0.0, 1.0).sort().values

    # ---- scene/motion.py:221-256
    def _sample_c2w_from_nu(self, idx, nu=None):
        if nu is None:
            nu = self._sample_nu_from_alignment(idx)
        elif torch.is_tensor(nu):
            nu = nu.to(self.device)
        else:
            raise NotImplementedError
        if self.curve_type == "quarternion_cartesian":
            rot_quaternion = self._rot(nu, idx)
            rot_quaternion = rot_quaternion / rot_quaternion.norm(dim=1, keepdim=True)
            return unitquat_to_rotmat(rot_quaternion), self._trans(nu, idx)
        se3 = torch.cat([self._trans(nu, idx), self._rot(nu, idx)], dim=1)
        c2w = se3_exp_map(se3)
        return c2w[:, :3, :3].transpose(-2, -1), c2w[:, 3, :3]

    def get_trajectory_matrices(self, idx, t=None, fused=None):
        """Batched _c2w_to_minicam: (world_view [K,4,4], full_proj [K,4,4], camera_center [K,3]).
        On device tensors the whole chain nu -> Bezier -> se3_exp_map -> cameras runs as one HIP kernel
        (pose.fused_trajectory); `fused=False` forces the torch-op reference implementation."""
        use_fused = (self.device.type == "cuda") if fused is None else fused
        if use_fused:   # both curve types (the rotation control points are [C+1,3] or [C+1,4])
            nu = self._sample_nu_from_alignment(idx) if t is None else t.to(self.device)
            if isinstance(idx, int):
                ct, cr = self._trans._control_points[idx], self._rot._control_points[idx]
            else:
                ct, cr = self._trans._control_points[idx][0], self._rot._control_points[idx][0]
            return fused_trajectory(ct, cr, nu, self.ref_cam.projection_matrix)
        rots, transes = self._sample_c2w_from_nu(idx, t)
        return c2w_to_view_proj(rots, transes, self.ref_cam.projection_matrix)

    def get_trajectory(self, idx, t=None):
        """List of MiniCam objects like the reference (scene/motion.py:162-178)."""
        wv, fp, cc = self.get_trajectory_matrices(idx, t)
        r = self.ref_cam
        return [MiniCam(r.image_width, r.image_height, r.FoVy, r.FoVx, r.znear, r.zfar, wv[i], fp[i], cc[i])
                for i in range(wv.shape[0])]

    def get_gt_image(self, idx):
        return self.gt_images[idx]

    def query(self, cam_idx: int, subframe_indice="all", post_process=None, background="random",
              compute_blurred=True, shard=None, uniform=None):
        """Render a blurry view (scene/motion.py:78-160).  Returns the reference's dict: 'blurred', 'gt',
        'subframes' [f,3,H,W], 'depths' [f,1,H,W], 'render_pkgs' (list of f per-subframe dicts whose
        'viewspace_points' entries are views of ONE [f,P,3] grad carrier, exposed as 'viewspace_points_all').

        shard=(rank, world): "subframes" sharding (deblurgs_amd.sharding): the view's f subframe cameras are computed on
        every rank (a few hundred flops), but only the slice [floor(rank f / world), floor((rank+1) f / world)) is
        rasterised here; 'subframes' / 'depths' / 'render_pkgs' then hold that slice, 'k0' its first index, 'K_total' = f,
        and 'blurred' is None (the blur needs the other ranks' subframes: sharding.subframe_sharded_loss_backward)."""
        assert self.gaussians is not None
        gaussians = self.gaussians
        if isinstance(background, str) and background == "random":
            bg = torch.rand(3, device=gaussians.get_xyz.device)
        else:
            bg = background
        if isinstance(subframe_indice, str) and subframe_indice == "all":
            nu = None if uniform is None else self._sample_nu_from_alignment(cam_idx, uniform)
        else:
            nu = self._sample_nu_from_alignment(cam_idx, uniform)
            if isinstance(subframe_indice, int):
                # NB: the reference's `== 1` special case is dead code (its result is overwritten,
                # scene/motion.py:129-131): 1 selects index 0, i.e. nu = 0.
                subfr_idx = torch.linspace(0, nu.shape[0] - 1, subframe_indice, device=nu.device).long()
            else:
                subfr_idx = subframe_indice
            nu = nu[subfr_idx]
        world_views, full_projs, centers = self.get_trajectory_matrices(cam_idx, nu)
        K_total, k0 = world_views.shape[0], 0
        if shard is not None:
            from .sharding import shard_range
            k0, k1 = shard_range(K_total, int(shard[0]), int(shard[1]))
            if k1 == k0:        # more ranks than subframes: nothing to rasterise here
                H, W = int(self.ref_cam.image_height), int(self.ref_cam.image_width)
                dev, P = world_views.device, gaussians.get_xyz.shape[0]
                e = lambda *shape, **kw: torch.zeros(shape, device=dev, **kw)
                return {"blurred": None, "gt": self.get_gt_image(cam_idx), "subframes": e(0, 3, H, W),
                        "depths": e(0, 1, H, W), "render_pkgs": [], "viewspace_points_all": e(0, P, 3),
                        "radii_all": e(0, P, dtype=torch.int32), "background": bg, "k0": k0, "K_total": K_total}
            world_views, full_projs, centers = world_views[k0:k1], full_projs[k0:k1], centers[k0:k1]
            compute_blurred = False
        pkg = gaussian_renderer.render_subframes(world_views, full_projs, centers, self.ref_cam, gaussians, bg)
        render_subframes = pkg["render"]
        # compute_blurred=False: the caller takes the blur from the fused loss kernel (losses.blur_l1_smooth)
        blurred = render_subframes.mean(dim=0) if compute_blurred else None
        if post_process is not None and blurred is not None:
            blurred = post_process(blurred)
        f = render_subframes.shape[0]
        render_pkgs = [{"render": render_subframes[i], "depth": pkg["depth"][i],
                        "viewspace_points": pkg["viewspace_points"],   # the shared [f,P,3] carrier
                        "subframe": i,
                        "visibility_filter": pkg["visibility_filter"][i], "radii": pkg["radii"][i]} for i in range(f)]
        return {"blurred": blurred, "gt": self.get_gt_image(cam_idx), "subframes": render_subframes,
                "depths": pkg["depth"], "render_pkgs": render_pkgs, "viewspace_points_all": pkg["viewspace_points"],
                "radii_all": pkg["radii"], "background": bg, "k0": k0, "K_total": K_total}
