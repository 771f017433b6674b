"""Densification-statistic consumers of the rasteriser outputs.

The reference walks the K per-subframe render packages in Python after every backward
(train.py:188-193 -> scene/gaussian_model.py:456-458): K x (masked max, masked norm-accumulate, masked add).
`add_densification_stats_subframes` does the same update for all K subframes with one kernel, reading the
[K,P,3] gradient of the fused operator's means2D carrier and its [K,P] radii.  The densifier itself (clone / split /
prune with the optimiser-state surgery) is deblurgs_amd.optim / GaussianCloud.densify_and_prune.
"""
import ctypes

import torch

from . import _lib


@torch.no_grad()
def add_densification_stats_subframes(viewspace_points, radii, max_radii2D, xyz_gradient_accum, denom, K_total=0,
                                      skip_flag_ptr=None):
    """In-place update of max_radii2D [P], xyz_gradient_accum [P,1] and denom [P,1] (float32, device tensors).
    K_total: number of subframes of the whole view when `radii` holds only this rank's share of them."""
    grad = viewspace_points.grad if viewspace_points.grad is not None else viewspace_points
    grad = grad.contiguous()
    K, P = radii.shape
    assert grad.shape == (K, P, 3) and radii.dtype == torch.int32
    for t in (max_radii2D, xyz_gradient_accum, denom):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.numel() == P
    st = ctypes.c_void_p(torch.cuda.current_stream(grad.device).cuda_stream)
    _lib.check(_lib.lib().dgs_densify_stats(grad.data_ptr(), radii.contiguous().data_ptr(), K, int(K_total), P,
                                            max_radii2D.data_ptr(), xyz_gradient_accum.data_ptr(), denom.data_ptr(),
                                            ctypes.c_void_p(skip_flag_ptr) if skip_flag_ptr else None, st),
               "dgs_densify_stats")
