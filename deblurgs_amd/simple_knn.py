"""distCUDA2 of the reference's simple-knn submodule (submodules/simple-knn/spatial.cu:15-26) on libdgs_hip.so:
mean squared distance of every point to its three nearest neighbours, used by create_from_pcd for the initial
scales (scene/gaussian_model.py:148-156).  `create_from_points` is that initialisation for a GaussianCloud."""
import ctypes

import numpy as np
import torch

from . import _lib

C0 = 0.28209479177387814   # utils/sh_utils.py:24


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 needs a HIP tensor (no CPU fallback)")
    pts = points.contiguous().float()
    P = pts.shape[0]
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)       # torch::full({P}, 0.0)
    L = _lib.lib()
    tmp = torch.empty(L.dgs_knn_tmp_bytes(P), dtype=torch.uint8, device=pts.device)
    st = ctypes.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)
    _lib.check(L.dgs_knn_mean_dist2(P, pts.data_ptr(), out.data_ptr(), tmp.data_ptr(), st), "dgs_knn_mean_dist2")
    return out


def create_from_points(points, colors, sh_degree=2, scale_lb=0.0, alpha_lower_bound=0.0, use_sigmoid=False,
                       device="cuda", **cloud_kw):
    """GaussianModel.create_from_pcd (scene/gaussian_model.py:138-168): SH dc = RGB2SH(colour), log-scale =
    log(sqrt(max(dist2, 1e-7)) - lb) on all three axes, identity rotation, opacity lb + (1 - lb) * 0.1."""
    from .cloud import GaussianCloud
    xyz = torch.as_tensor(np.asarray(points), dtype=torch.float32, device=device)
    col = torch.as_tensor(np.asarray(colors), dtype=torch.float32, device=device)
    if use_sigmoid:
        col = torch.log(col / (1 - col))
        fused_color = col / C0                      # utils/sh_utils.py RGB2SH(use_sigmoid=True)
    else:
        fused_color = (col - 0.5) / C0
    P = xyz.shape[0]
    M = (sh_degree + 1) ** 2
    features = torch.zeros((P, 3, M), device=device)
    features[:, :3, 0] = fused_color
    dist2 = torch.clamp_min(distCUDA2(xyz), 0.0000001)
    scales = torch.log((torch.sqrt(dist2) - scale_lb).clamp_min(0.001))[..., None].repeat(1, 3)   # LowerBoundLog
    rots = torch.zeros((P, 4), device=device)
    rots[:, 0] = 1
    lb = alpha_lower_bound
    opac = (lb + (1.0 - lb) * (0.1 * torch.ones((P, 1), device=device))).clamp(0.0, 1.0)
    return GaussianCloud(xyz, features[:, :, 0:1].transpose(1, 2).contiguous(), features[:, :, 1:].transpose(1, 2).contiguous(),
                         scales, rots, opac, sh_degree=sh_degree, active_sh_degree=0, scale_lb=scale_lb,
                         alpha_lower_bound=alpha_lower_bound, use_sigmoid=use_sigmoid, **cloud_kw)
