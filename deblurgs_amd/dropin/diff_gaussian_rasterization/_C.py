"""`diff_gaussian_rasterization._C` on the MI355X C ABI: the reference's own L0 entry points, same positional
signatures, same return tuples.

The reference's pybind module (submodules/diff-gaussian-rasterization/ext.cpp:15-19) exports

    rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, z_near, z_far, image_height, image_width, sh,
                        degree, campos, prefiltered, use_sigmoid, debug)
        -> (num_rendered, out_color [3,H,W], out_depth [1,H,W], radii [P], geomBuffer, binningBuffer, imgBuffer)
    rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, z_near, z_far, dL_dout_color, dL_dout_depth, sh,
                        degree, campos, geomBuffer, R, binningBuffer, imageBuffer, use_sigmoid, debug)
        -> (dL_dmeans2D [P,3], dL_dcolors [P,3], dL_dopacity [P,1], dL_dmeans3D [P,3], dL_dcov3D [P,6],
            dL_dsh [P,M,3], dL_dscales [P,3], dL_drotations [P,4], dL_dviewmatrix [4,4], dL_dprojmatrix [4,4])
    mark_visible(means3D, viewmatrix, projmatrix) -> bool [P]

(rasterize_points.h:18-73, rasterize_points.cu:35-239).  With this file in place of the CUDA extension the reference's
UNMODIFIED diff_gaussian_rasterization/__init__.py (:66-101 forward, :120-160 backward) runs on libdgs_hip.so: an absent
input is an empty tensor as GaussianRasterizer.forward passes it (:222-233), every gradient comes back as a tensor --
zeros where the reference's zero-initialised tensor is never written (dL_dscales / dL_drotations with cov3D_precomp,
dL_dsh [P,0,3] with colours, everything when P == 0: rasterize_points.cu:162-176) --, the three buffers are opaque byte
tensors the caller keeps between the two calls.  K = 1 of the fused operator: same kernels, same bits as
deblurgs_amd.diff_gaussian_rasterization (tests/test_gpu_parity.py::test_l0_C_module_*).

num_rendered is an int (an int subclass that remembers which duplicate rule built the buffers; a plain int -- a caller
that stored it as such -- means the package default, deblurgs_amd.diff_gaussian_rasterization.TILE_CULL).
"""
import ctypes
from collections import namedtuple

import torch

from deblurgs_amd import _lib
from deblurgs_amd import diff_gaussian_rasterization as _dgr

__all__ = ["rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"]

# the settings fields _forward_impl / _backward_impl read (GaussianRasterizationSettings, __init__.py:172-187)
_Settings = namedtuple("_Settings", "image_height image_width tanfovx tanfovy bg scale_modifier z_near z_far use_sigmoid "
                                    "sh_degree campos prefiltered debug")


def _settings(bg, scale_modifier, tan_fovx, tan_fovy, z_near, z_far, H, W, degree, campos, prefiltered, use_sigmoid,
              debug):
    return _Settings(int(H), int(W), float(tan_fovx), float(tan_fovy), bg, float(scale_modifier), float(z_near),
                     float(z_far), bool(use_sigmoid), int(degree), campos, bool(prefiltered), bool(debug))


def _inputs(means3D, colors, scales, rotations, cov3D_precomp, sh, viewmatrix, projmatrix, campos):
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")          # rasterize_points.cu:60-62
    f, o = _dgr._f32c, _dgr._opt
    return (f(means3D), f(o(sh)), f(o(colors)), f(o(scales)), f(o(rotations)), f(o(cov3D_precomp)),
            f(viewmatrix).reshape(1, 4, 4), f(projmatrix).reshape(1, 4, 4), f(campos.to(means3D.device)).reshape(1, 3))


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, z_near, z_far, image_height, image_width, sh, degree,
                        campos, prefiltered, use_sigmoid, debug):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-123)."""
    m3, shc, colc, scc, rotc, covc, viewm, projm, cam = _inputs(means3D, colors, scales, rotations, cov3D_precomp, sh,
                                                                viewmatrix, projmatrix, campos)
    rs = _settings(background, scale_modifier, tan_fovx, tan_fovy, z_near, z_far, image_height, image_width, degree,
                   campos, prefiltered, use_sigmoid, debug)
    with torch.no_grad():
        R, color, depth, radii, geom, binning, img = _dgr._forward_impl(
            1, m3, shc, colc, _dgr._f32c(opacity), scc, rotc, covc, viewm, projm, cam, rs)
    return R, color[0], depth[0], radii[0], geom, binning, img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, z_near, z_far, dL_dout_color, dL_dout_depth,
                                 sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, use_sigmoid, debug):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:125-218).  H and W come from dL_dout_color, as there (:154-155)."""
    m3, shc, colc, scc, rotc, covc, viewm, projm, cam = _inputs(means3D, colors, scales, rotations, cov3D_precomp, sh,
                                                                viewmatrix, projmatrix, campos)
    P = m3.shape[0]
    H, W = int(dL_dout_color.shape[-2]), int(dL_dout_color.shape[-1])
    rs = _settings(background, scale_modifier, tan_fovx, tan_fovy, z_near, z_far, H, W, degree, campos, False, use_sigmoid,
                   debug)
    if not isinstance(R, _dgr._NumRendered):       # a caller that kept the count as a plain int: the package's duplicate rule
        n = _dgr._NumRendered(int(R))
        n.tile_cull = bool(_dgr.TILE_CULL)
        R = n
    gd = _dgr._f32c(_dgr._opt(dL_dout_depth))
    with torch.no_grad():
        (g_means2D, g_colors, g_opacity, g_means3D, g_cov3D, g_sh, g_scales, g_rots, g_view, g_proj) = _dgr._backward_impl(
            1, R, m3, shc, colc, (P, 1), scc, rotc, covc, viewm, projm, cam, rs, radii.reshape(1, P).contiguous(),
            geomBuffer, binningBuffer, imageBuffer, _dgr._f32c(dL_dout_color).reshape(1, 3, H, W),
            None if gd is None else gd.reshape(1, 1, H, W))
    z = dict(dtype=torch.float32, device=m3.device)
    # the reference hands back zero-initialised tensors for what its kernels never write (rasterize_points.cu:162-176)
    if g_sh is None:
        g_sh = torch.zeros((P, 0, 3), **z)
    if g_scales is None:
        g_scales = torch.zeros((P, 3), **z)
    if g_rots is None:
        g_rots = torch.zeros((P, 4), **z)
    return (g_means2D[0], g_colors, g_opacity, g_means3D, g_cov3D, g_sh, g_scales, g_rots, g_view[0], g_proj[0])


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:220-239)."""
    with torch.no_grad():
        m3, vm, pm = _dgr._f32c(means3D), _dgr._f32c(viewmatrix), _dgr._f32c(projmatrix)
        present = torch.zeros(m3.shape[0], dtype=torch.bool, device=m3.device)
        if m3.shape[0] != 0:
            _lib.check(_lib.lib().dgs_mark_visible(m3.shape[0], _dgr._ptr(m3), _dgr._ptr(vm), _dgr._ptr(pm),
                                                   ctypes.c_void_p(present.data_ptr()), _dgr._stream(m3.device)),
                       "dgs_mark_visible")
    return present
