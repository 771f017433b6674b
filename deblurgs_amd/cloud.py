"""Parameter container exposing exactly the getters the rasteriser adapter reads from the reference's
GaussianModel (scene/gaussian_model.py:114-137) with the reference's activations
(scene/gaussian_activation.py:29-52, scene/gaussian_model.py:36-50): opacity clamp(0,1), scale exp(x)+lb,
rotation F.normalize, SH = cat(dc, rest).  The optimiser / densifier around it are out of scope.
"""
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
                 z_near=0.2, z_far=100.0, use_sigmoid=False, scale_lb=0.0):
        super().__init__()
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree if active_sh_degree is None else active_sh_degree
        self.z_near = z_near
        self.z_far = z_far
        self.use_sigmoid = use_sigmoid
        self._xyz = nn.Parameter(xyz)
        self._features_dc = nn.Parameter(features_dc)      # [P,1,3]
        self._features_rest = nn.Parameter(features_rest)  # [P,M-1,3]
        self._scaling = nn.Parameter(scaling)              # log-scale
        self._rotation = nn.Parameter(rotation)
        self._opacity = nn.Parameter(opacity)              # identity-with-clamp activation
        self.scaling_activation = LowerBoundExponent(scale_lb)
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

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def hot_parameters(self):
        """The per-Gaussian tensors whose gradients the sharded loop all-reduces."""
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]
