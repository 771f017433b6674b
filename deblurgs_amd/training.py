"""One training iteration of the reference (train.py:104-208) on the fused path: scheduled hyper-parameters ->
CameraMotionModule.query (K subframes, one launch chain) -> fused blur loss + hinge -> backward -> densification
statistics (one kernel) -> densify_and_prune / reset_opacity on their schedule -> one fused Adam launch for the
Gaussians and the trajectory.  This is the caller-side glue of SURVEY 8f rows f1-f3, not a replacement of train.py's
dataset / logging / checkpoint code."""
import types

import torch

from . import losses
from .cloud import get_expon_lr_func
from .densify_stats import add_densification_stats_subframes


def default_optimization_params(**overrides):
    """arguments/__init__.py:84-123 (OptimizationParams defaults)."""
    d = dict(iterations=150_000, position_lr_init=0.00016, position_lr_final=0.0000016, feature_lr=0.0025,
             opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, percent_dense=0.01, lambda_t_smooth_init=1e-3,
             lambda_t_smooth_final=1e-5, lambda_depth_tv=0.0, lambda_hinge=0.1, densification_interval=200,
             opacity_reset_interval=3000, densify_from_iter=500, densify_until_iter=75_000,
             densify_grad_threshold_init=4e-4, densify_grad_threshold_final=2e-4, densify_annealing_until=25_000,
             clip_grad=-1.0, curve_controlpoints_lr=1e-2, curve_rotation_lr=1e-3, curve_alignment_lr=0.0,
             curve_lr_half_iter=15_000, curve_start_iter=1000, curve_end_iter=100_000)
    d.update(overrides)
    return types.SimpleNamespace(**d)


class TrainingLoop:
    def __init__(self, gaussians, cam_motion_module, opt, cameras_extent, white_background=False, spatial_lr_scale=None,
                 distributed=False):
        """distributed=True ("views" sharding, deblurgs_amd.sharding): every rank steps on its own view; the
        per-Gaussian gradients are averaged over ranks before the Adam step and the densification statistics are
        combined before every densify_and_prune, so the replicas stay identical."""
        self.gaussians, self.motion, self.opt, self.extent = gaussians, cam_motion_module, opt, cameras_extent
        self.distributed = distributed
        self._stat_prev = None
        self.white_background = white_background
        gaussians.training_setup(opt, spatial_lr_scale=cameras_extent if spatial_lr_scale is None else spatial_lr_scale)
        cam_motion_module.link_gaussian(gaussians)
        cam_motion_module.add_training_setup(gaussians, {"curve_rot": opt.curve_rotation_lr,
                                                         "curve_trans": opt.curve_controlpoints_lr,
                                                         "curve_alignment": opt.curve_alignment_lr})
        self.densify_threshold_func = get_expon_lr_func(opt.densify_grad_threshold_init, opt.densify_grad_threshold_final,
                                                        max_steps=opt.densify_annealing_until)
        self.lambda_t_smooth_func = get_expon_lr_func(opt.lambda_t_smooth_init, opt.lambda_t_smooth_final,
                                                      max_steps=opt.iterations)
        if opt.curve_start_iter > 1:
            cam_motion_module.alternate_optimization()          # train.py:100 -- curve gradients off until curve_start_iter

    def step(self, iteration, cam_idx):
        g, opt = self.gaussians, self.opt
        g.update_learning_rate(iteration, opt)
        densification_threshold = self.densify_threshold_func(iteration)
        lambda_t_smooth = self.lambda_t_smooth_func(iteration)
        if iteration == opt.curve_start_iter or iteration == opt.curve_end_iter:
            self.motion.alternate_optimization()
        if iteration % 1000 == 0:
            g.oneupSHdegree()
        subframe_indice = "all" if iteration >= opt.curve_start_iter else 1
        # The opacity hinge is built BEFORE the render: autograd then runs its backward AFTER the rasteriser's, so the
        # rasteriser's gradient (a view of the flat gradient buffer) becomes `_opacity.grad` and the hinge term is added
        # into it in place -- the six gradients stay one contiguous bucket for the all-reduce.
        L_hinge = losses.hinge_l2(g._opacity) if opt.lambda_hinge > 0.0 else None
        r = self.motion.query(cam_idx=cam_idx, subframe_indice=subframe_indice, compute_blurred=False)
        total, blur, lv = losses.blur_l1_smooth(r["subframes"], r["gt"], lambda_t_smooth)
        Ll1, L_t = lv[0], lv[1]
        loss = total
        if L_hinge is not None:
            loss = loss + opt.lambda_hinge * L_hinge
        if opt.lambda_depth_tv > 0.0:
            loss = loss + opt.lambda_depth_tv * losses.tv_loss(r["depths"])
        loss.backward()
        if self.distributed:
            from . import sharding
            sharding.flat_allreduce_grads(g.hot_parameters(), average=True)
        with torch.no_grad():
            if iteration < opt.densify_until_iter:
                if self.distributed and self._stat_prev is None:
                    self._stat_prev = (g.xyz_gradient_accum.clone(), g.denom.clone())
                add_densification_stats_subframes(r["viewspace_points_all"], r["radii_all"], g.max_radii2D,
                                                  g.xyz_gradient_accum, g.denom)
                if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                    if self.distributed:
                        sharding.allreduce_densification_stats(g, self._stat_prev)
                        self._stat_prev = None          # densify_and_prune resets the statistics to zeros
                    gen = None
                    if self.distributed:                # the split's normal draws must be the same on every rank
                        gen = torch.Generator(device=g._xyz.device)
                        gen.manual_seed(1_000_003 * int(iteration) + 17)
                    g.densify_and_prune(densification_threshold, self.extent, generator=gen)
                if iteration % opt.opacity_reset_interval == 0 or (self.white_background and
                                                                   iteration == opt.densify_from_iter):
                    g.reset_opacity()
            if iteration < opt.iterations:
                g.optimizer.step()                      # clip_grad_value_ is fused into the step (FusedAdam.clip_value)
                g.optimizer.zero_grad(set_to_none=True)
        return {"loss": loss.detach(), "l1": Ll1, "smooth": L_t, "hinge": L_hinge, "num_points": g._xyz.shape[0]}
