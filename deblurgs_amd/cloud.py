"""Parameter container exposing exactly the getters the rasteriser adapter reads from the reference's
GaussianModel (scene/gaussian_model.py:114-137) with the reference's activations
(scene/gaussian_activation.py:29-52, scene/gaussian_model.py:36-50): opacity clamp(0,1), scale exp(x)+lb,
rotation F.normalize, SH = cat(dc, rest), plus the training-side methods of GaussianModel that follow the hot path
in an iteration: training_setup / update_learning_rate (:170-204), densify_and_prune (:436-448),
prune_points (:336-349), reset_opacity (:247-253), backed by the fused kernels of deblurgs_amd.optim.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class Clamp(nn.Module):
    def forward(self, x):
        return x.clamp(0.0, 1.0)


class LowerBoundExponent(nn.Module):
    def __init__(self, lower_bound):
        super().__init__()
        self.lower_bound = lower_bound

    def forward(self, x):
        return torch.exp(x) + self.lower_bound


class GaussianCloud(nn.Module):
    def __init__(self, xyz, features_dc, features_rest, scaling, rotation, opacity, sh_degree=2, active_sh_degree=None,
                 z_near=0.2, z_far=100.0, use_sigmoid=False, scale_lb=0.0, alpha_lower_bound=0.0, use_isotrophic=False):
        super().__init__()
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree if active_sh_degree is None else active_sh_degree
        self.z_near = z_near
        self.z_far = z_far
        self.use_sigmoid = use_sigmoid
        # one shared scale per Gaussian: column 0 of _scaling, which stays [P,3] (scene/gaussian_model.py:75,115-118;
        # the spelling is the reference's)
        self.use_isotrophic = bool(use_isotrophic)
        self._xyz = nn.Parameter(xyz)
        self._features_dc = nn.Parameter(features_dc)      # [P,1,3]
        self._features_rest = nn.Parameter(features_rest)  # [P,M-1,3]
        self._scaling = nn.Parameter(scaling)              # log-scale
        self._rotation = nn.Parameter(rotation)
        self._opacity = nn.Parameter(opacity)              # identity-with-clamp activation
        self.scaling_activation = LowerBoundExponent(scale_lb)
        self.scale_lower_bound = scale_lb
        # render_subframes() hands the raw parameters to the kernels, which apply exactly the activations below
        # (set False to go through the getters, as the reference's render() does)
        self.fused_activations = True
        self.alpha_lower_bound = alpha_lower_bound
        self.optimizer = None
        self.percent_dense = 0.0
        self.spatial_lr_scale = 1.0
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.opacity_activation = Clamp()
        self.rotation_activation = F.normalize

    @classmethod
    def from_scene(cls, scene, device="cuda"):
        """Build from a deblurgs_amd.synthetic scene dict (activated values -> raw parameters)."""
        t = lambda a: torch.from_numpy(a).to(device)
        sh = t(scene["sh"])
        return cls(t(scene["means3D"]), sh[:, :1].contiguous(), sh[:, 1:].contiguous(),
                   torch.log(t(scene["scales"])), t(scene["rotations"]), t(scene["opacities"]),
                   sh_degree=scene["sh_degree"], z_near=scene["z_near"], z_far=scene["z_far"])

    @property
    def get_scaling(self):
        if self.use_isotrophic:
            return self.scaling_activation(self._scaling[:, :1].expand(-1, 3))
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @torch.no_grad()
    def device_activations(self):
        """(get_scaling, get_rotation, get_opacity) as the raw-parameter kernels evaluate them (dgs_cloud_activations):
        bit-identical to what render_subframes() rasterises with fused_activations; the torch getters above can differ
        from it by an ulp of exp()."""
        import ctypes
        from . import _lib
        dev = self._xyz.device
        P = self._xyz.shape[0]
        raw_sc = self._scaling[:, :1].expand(-1, 3) if self.use_isotrophic else self._scaling
        sc, rot, op = (t.detach().float().contiguous() for t in (raw_sc, self._rotation, self._opacity))
        o_sc, o_rot, o_op = torch.empty_like(sc), torch.empty_like(rot), torch.empty_like(op)
        _lib.check(_lib.lib().dgs_cloud_activations(P, sc.data_ptr(), rot.data_ptr(), op.data_ptr(),
                                                    float(self.scale_lower_bound), o_sc.data_ptr(), o_rot.data_ptr(),
                                                    o_op.data_ptr(),
                                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "dgs_cloud_activations")
        return o_sc, o_rot, o_op

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def hot_parameters(self):
        """The per-Gaussian tensors whose gradients the sharded loop all-reduces."""
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    # ------------------------------------------------------------------ checkpoints (scene/gaussian_model.py:80-112)
    def capture(self):
        """The reference's checkpoint tuple, field for field (train.py:214-216 saves (capture(), iteration))."""
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(),
                self.spatial_lr_scale)

    def restore(self, model_args, training_args, fused=True, cam_motion_module=None, curve_lrs=None):
        """scene/gaussian_model.py:97-112.  The checkpoint's optimiser state may hold the three trajectory groups that
        CameraMotionModule.add_training_setup appended (curve_rot / curve_trans / curve_alignment, scene/motion.py:63-76)
        after the six per-Gaussian ones; the reference's own restore() then fails in load_state_dict (6 fresh groups vs
        9 saved).  Here: with `cam_motion_module` given its groups are attached first and everything is loaded; without
        it only the per-Gaussian groups and their moments are loaded."""
        (self.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, self.max_radii2D, xyz_gradient_accum,
         denom, opt_dict, spatial_lr_scale) = model_args
        dev = self._xyz.device
        as_param = lambda t: nn.Parameter(t.detach().to(dev).float().contiguous().requires_grad_(True))
        (self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
         self._opacity) = (as_param(t) for t in (xyz, f_dc, f_rest, scaling, rotation, opacity))
        self.training_setup(training_args, spatial_lr_scale=spatial_lr_scale, fused=fused)
        self.max_radii2D = self.max_radii2D.to(dev)
        self.xyz_gradient_accum = xyz_gradient_accum.to(dev)
        self.denom = denom.to(dev)
        saved_groups = [dict(g) for g in opt_dict["param_groups"]]       # never edit the caller's checkpoint in place
        opt_dict = {"state": opt_dict["state"], "param_groups": saved_groups}
        curve = [g for g in saved_groups if str(g.get("name", "")).startswith("curve_")]
        if curve and cam_motion_module is not None:
            lrs = {g["name"]: g["lr"] for g in curve}
            lrs.update(curve_lrs or {})
            cam_motion_module.add_training_setup(self, {"curve_rot": lrs.get("curve_rot", 0.0),
                                                        "curve_trans": lrs.get("curve_trans", 0.0),
                                                        "curve_alignment": lrs.get("curve_alignment", 0.0)})
        elif curve:
            keep = [g for g in saved_groups if not str(g.get("name", "")).startswith("curve_")]
            ids = {i for g in keep for i in g["params"]}
            opt_dict = {"state": {k: v for k, v in opt_dict["state"].items() if k in ids}, "param_groups": keep}
        # group keys a foreign Adam did not write (torch versions differ) take this optimiser's defaults
        for g in opt_dict["param_groups"]:
            for key, val in self.optimizer.defaults.items():
                g.setdefault(key, val)
        self.optimizer.load_state_dict(opt_dict)

    # ------------------------------------------------------------------ training side
    def _named(self):
        return dict(zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), self.hot_parameters()))

    def _set_named(self, tensors):
        (self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling,
         self._rotation) = (tensors[n] for n in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"))

    def training_setup(self, training_args, spatial_lr_scale=1.0, fused=True):
        """scene/gaussian_model.py:170-195.  training_args needs percent_dense, position_lr_init/final, feature_lr,
        opacity_lr, scaling_lr, rotation_lr, iterations (and optionally clip_grad)."""
        from .optim import FusedAdam
        self.spatial_lr_scale = spatial_lr_scale
        self.percent_dense = training_args.percent_dense
        dev = self._xyz.device
        P = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)
        a = training_args
        l = [
            {'params': [self._xyz], 'lr': a.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {'params': [self._features_dc], 'lr': a.feature_lr, "name": "f_dc"},
            {'params': [self._features_rest], 'lr': a.feature_lr / 20.0, "name": "f_rest"},
            {'params': [self._opacity], 'lr': a.opacity_lr, "name": "opacity"},
            {'params': [self._scaling], 'lr': a.scaling_lr, "name": "scaling"},
            {'params': [self._rotation], 'lr': a.rotation_lr, "name": "rotation"},
        ]
        if fused:
            self.optimizer = FusedAdam(l, lr=0.0, eps=1e-15, clip_value=getattr(a, "clip_grad", 0.0))
        else:
            self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=a.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=a.position_lr_final * self.spatial_lr_scale,
                                                    max_steps=a.iterations)

    def update_learning_rate(self, iteration, opt=None, alignment_lr=None):
        """scene/gaussian_model.py:197-210: xyz follows the exponential schedule, the curve groups halve every
        curve_lr_half_iter iterations once the curve optimisation has started, curve_alignment is set from outside."""
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group["lr"] = self.xyz_scheduler_args(iteration)
            elif opt is not None and group["name"] in ["curve_rot", "curve_trans"] and iteration >= opt.curve_start_iter:
                group["lr"] = group["lr"] * (0.5) ** (1 / opt.curve_lr_half_iter)
            elif alignment_lr is not None and group["name"] == "curve_alignment":
                group["lr"] = alignment_lr

    def _moments(self):
        m, v, steps = [], [], []
        named = self._named()
        for n, p in named.items():
            st = self.optimizer.state.get(p, None) if self.optimizer is not None else None
            st = st if st else None
            m.append(None if st is None else st["exp_avg"])
            v.append(None if st is None else st["exp_avg_sq"])
            steps.append(st)
        return m, v, steps

    def _install(self, new_p, new_m, new_v, old_states):
        """Swap the parameters (and their optimiser state) for new tensors, as _prune_optimizer /
        cat_tensors_to_optimizer do (:315-334, :359-387): the state dict object, hence `step`, is kept."""
        names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
        old = self._named()
        tensors = {}
        for i, n in enumerate(names):
            tensors[n] = nn.Parameter(new_p[i].requires_grad_(True))
        if self.optimizer is not None:
            for group in self.optimizer.param_groups:
                n = group.get("name")
                if n not in tensors:
                    continue
                i = names.index(n)
                st = old_states[i]
                if old[n] in self.optimizer.state:
                    del self.optimizer.state[old[n]]
                group["params"][0] = tensors[n]
                if st is not None:
                    st["exp_avg"], st["exp_avg_sq"] = new_m[i], new_v[i]
                    self.optimizer.state[tensors[n]] = st
        self._set_named(tensors)

    @torch.no_grad()
    def densify_and_prune(self, max_grad, extent, noise=None, generator=None):
        """scene/gaussian_model.py:436-448 in three kernels + four scans.  `noise` ([2 m, 3] standard normals for
        the m split-selected Gaussians x 2 copies) is drawn here when absent (the reference's torch.normal); a callable is
        called with m and returns them."""
        from . import optim
        min_opacity = self.alpha_lower_bound + (1 - self.alpha_lower_bound) * 0.005
        counts, flags, offs = optim.densify_plan(self.xyz_gradient_accum, self.denom, self._scaling.detach(),
                                                 self._opacity.detach(), max_grad, self.percent_dense * extent,
                                                 min_opacity, self.scale_lower_bound, isotropic=self.use_isotrophic)
        if callable(noise):      # (tests: the same normal draws as another implementation of the same step)
            noise = torch.as_tensor(noise(counts[3]), dtype=torch.float32, device=self._xyz.device)
        if noise is None:
            noise = torch.randn((2 * counts[3], 3), device=self._xyz.device, generator=generator)
        m, v, states = self._moments()
        params = [p.detach().contiguous() for p in self.hot_parameters()]
        new_p, new_m, new_v = optim.densify_apply(counts, flags, offs, params, m, v, noise, self.scale_lower_bound,
                                                  isotropic=self.use_isotrophic)
        self._install(new_p, new_m, new_v, states)
        dev = self._xyz.device
        Pn = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((Pn, 1), device=dev)
        self.denom = torch.zeros((Pn, 1), device=dev)
        self.max_radii2D = torch.zeros((Pn,), device=dev)
        return counts

    @torch.no_grad()
    def prune_points(self, mask):
        """scene/gaussian_model.py:336-349 (plain torch indexing: used outside the training loop)."""
        valid = ~mask
        m, v, states = self._moments()
        new_p = [p.detach()[valid].contiguous() for p in self.hot_parameters()]
        new_m = [None if t is None else t[valid].contiguous() for t in m]
        new_v = [None if t is None else t[valid].contiguous() for t in v]
        self._install(new_p, new_m, new_v, states)
        self.xyz_gradient_accum = self.xyz_gradient_accum[valid]
        self.denom = self.denom[valid]
        self.max_radii2D = self.max_radii2D[valid]

    @torch.no_grad()
    def reset_opacity(self, new_opacity=None):
        """scene/gaussian_model.py:247-253 + replace_tensor_to_optimizer (:301-313): moments are zeroed."""
        if new_opacity is None:
            lb = self.alpha_lower_bound
            new_opacity = lb + (1 - lb) * float(self.opacity_activation(torch.ones(1) * 0.1))
        opacities_new = torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * new_opacity).clamp(0.0, 1.0)
        old = self._opacity
        new = nn.Parameter(opacities_new.detach().clone().requires_grad_(True))
        if self.optimizer is not None:
            for group in self.optimizer.param_groups:
                if group.get("name") == "opacity":
                    st = self.optimizer.state.get(old, None)
                    if old in self.optimizer.state:
                        del self.optimizer.state[old]
                    group["params"][0] = new
                    if st:
                        st["exp_avg"] = torch.zeros_like(new)
                        st["exp_avg_sq"] = torch.zeros_like(new)
                        self.optimizer.state[new] = st
        self._opacity = new


def get_scheduler(lr_init, lr_final, warmup_ratio, step_warmup, step_final):
    """utils/general_utils.py:72-101 as the fork left it: 0 up to and including step_warmup (the exponential warm-up is
    commented out there, so warmup_ratio has no effect), exponential decay lr_init -> lr_final over
    (step_warmup, step_final], lr_final afterwards; identically 0 for lr_init <= 1e-8 inside the decay phase.  Drives
    the curve_alignment learning rate (train.py:90-94,109)."""
    import math

    def get_lr(step):
        if step < 1:
            raise ValueError("Step must be greater than 0")
        if step <= step_warmup:
            return 0.0
        if step <= step_final:
            if lr_init <= 1e-8:
                return 0.0
            decay_rate = math.log(lr_final / lr_init) / (step_final - step_warmup)
            return lr_init * math.exp(decay_rate * (step - step_warmup))
        return lr_final

    return get_lr


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, max_steps=1000000):
    """utils/general_utils.py:31-71 (the fork's variant: clamps outside [0, max_steps], constant when
    lr_init <= lr_final, log-linear interpolation otherwise).  Also drives the densification-threshold annealing
    (train.py:79-81,110)."""
    state = dict(lr_final=lr_final)

    def helper(step):
        s = step - lr_delay_steps
        ms = max_steps - lr_delay_steps
        if s < 0:
            return lr_init
        if s > ms:
            return state["lr_final"]
        if lr_init <= 0.0:
            return 0.0
        if lr_init <= state["lr_final"]:
            return lr_init
        if state["lr_final"] <= 0.0:
            state["lr_final"] = 1e-6
        t = np.clip(s / ms, 0, 1)
        return float(np.exp(np.log(lr_init) * (1 - t) + np.log(state["lr_final"]) * t))

    return helper
