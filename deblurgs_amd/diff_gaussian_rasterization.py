"""Operator surface of the rasteriser -- same names, argument order, return arity, dtypes and error behaviour
as the reference's /root/reference/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py
(GaussianRasterizationSettings :172-187, GaussianRasterizer :189-241, rasterize_gaussians :21-46,
_RasterizeGaussians :48-170), backed by libdgs_hip.so through ctypes instead of the pybind `_C` module.

Additions for the blur-integration loop: `_RasterizeGaussiansK` / `rasterize_gaussians_subframes` /
`GaussianRasterizer.forward_subframes` rasterise all K subframe poses of one blurry view in ONE fused launch
chain (the reference calls the K=1 operator K times from scene/motion.py:141-143).
"""
import ctypes
import os
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    z_near: float
    z_far: float
    use_sigmoid: bool
    sh_degree: int
    campos: torch.Tensor      # [3]; [K,3] for the fused K-subframe operator
    prefiltered: bool
    debug: bool


# Tile culling (DgsProblem.tile_cull, include/dgs_hip.h): drop, at duplication time, the (tile, Gaussian) pairs the
# reference would skip at every pixel of the tile.  Outputs are unchanged; TILE_CULL = False (or DGS_TILE_CULL=0)
# reproduces the reference's rectangle lists bit for bit.
TILE_CULL = os.environ.get("DGS_TILE_CULL", "1") != "0"
# DgsProblem.wide_records: True keeps key + value arrays for the duplicates even when the one-word record fits (the tests
# of that storage); carried by every problem this module builds, forward and backward alike.
WIDE_RECORDS = False
# Test hook: a dict placed here receives the backward's scratch blob and internal gradients ("scratch", "R", "K", "P",
# "dL_dcov3D", "dL_dcolors") so that the parity tests can read the compositing backward's per-(subframe, Gaussian) totals
# (dgs_backward_scratch_layout) -- dL_dconic / dL_dopacity / dL_dcov3D before the ill-conditioned scale / rotation chain.
BACKWARD_DEBUG = None


class _NumRendered(int):
    """num_rendered as the reference returns it, remembering which duplicate rule produced the state blobs."""
    tile_cull = False


# ---------------------------------------------------------------------------------------------- plumbing
_pinned = {}


def _pinned_word(device):
    key = (device.type, device.index)
    if key not in _pinned:
        _pinned[key] = torch.zeros(8, dtype=torch.int32).pin_memory()
    return _pinned[key]


def _opt(t):
    """The reference passes torch.Tensor([]) for an absent input; map absent/empty to None."""
    if t is None or t.numel() == 0:
        return None
    return t


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _State:
    """What backward needs (the reference's geomBuffer / binningBuffer / imgBuffer + num_rendered)."""
    __slots__ = ("R", "K", "P", "M", "W", "H")


def _make_problem(K, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewm, projm, campos,
                  rs, geom, image, binning, tile_cull, raw=None):
    p = _lib.DgsProblem()
    p.context = _lib.context(means3D.device.index)   # the package's per-device context (side stream, stage timers)
    p.tile_cull = int(bool(tile_cull))
    p.wide_records = int(bool(WIDE_RECORDS))
    p.raw_params = 0 if raw is None else (3 if raw.get("isotropic") else 1)
    p.scale_lb = 0.0 if raw is None else float(raw["scale_lb"])
    p.shs_rest = None if raw is None else _ptr(raw["sh_rest"])
    p.P = means3D.shape[0]
    p.D = int(rs.sh_degree)
    p.M = 0 if sh is None else sh.shape[1]
    p.W = int(rs.image_width)
    p.H = int(rs.image_height)
    p.K = K
    p.tanfovx = float(rs.tanfovx)
    p.tanfovy = float(rs.tanfovy)
    p.scale_modifier = float(rs.scale_modifier)
    p.z_near = float(rs.z_near)
    p.z_far = float(rs.z_far)
    p.use_sigmoid = int(bool(rs.use_sigmoid))
    p.prefiltered = int(bool(rs.prefiltered))
    p.debug = int(bool(rs.debug))
    p.means3D = _ptr(means3D)
    p.shs = _ptr(sh)
    p.colors_precomp = _ptr(colors_precomp)
    p.opacities = _ptr(opacities)
    p.scales = _ptr(scales)
    p.rotations = _ptr(rotations)
    p.cov3D_precomp = _ptr(cov3D_precomp)
    p.viewmatrix = _ptr(viewm)
    p.projmatrix = _ptr(projm)
    p.campos = _ptr(campos)
    p.bg = _ptr(rs._bg_c)
    p.geom_state = _ptr(geom)
    p.geom_bytes = 0 if geom is None else geom.numel()
    p.image_state = _ptr(image)
    p.image_bytes = 0 if image is None else image.numel()
    p.binning_state = _ptr(binning)
    p.binning_bytes = 0 if binning is None else binning.numel()
    return p


class _RS:
    """Settings with contiguous fp32 device copies of bg (kept alive for the call)."""

    def __init__(self, rs, device):
        self.__dict__.update(rs._asdict())
        self._bg_c = _f32c(rs.bg.to(device))


def _forward_impl(K, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewm, projm, campos,
                  raster_settings, raw=None, capacity=None, forward_only=False, debug_checksum=False):
    """raw = {"scale_lb": float, "sh_rest": [P,M-1,3] or None, "isotropic": bool}: the inputs are the cloud's raw parameters
    (DgsProblem.raw_params) and sh is the dc part [P,1,3].
    capacity: size the duplicate arrays for that many duplicates up front and run the one-call dgs_forward (no host
    read between the phases; what fused_step.FusedStep does every iteration).  The returned count then carries
    `.capacity` (the binning blob is laid out for it) and `.overflow`.
    forward_only: an inference call (DgsProblem.forward_only): nothing is kept for a backward -- the image blob holds the
    tile ranges alone, final_T / n_contrib / cov3D / the activation mask are not stored.
    debug_checksum (parity tests; tile_cull off): the returned count carries `.contrib_checksum`, an int32 [K, H*W] tensor
    of DgsForwardOut.debug_contrib_checksum -- which pairs contributed to each pixel."""
    L = _lib.lib()
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:60-62
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("deblurgs_amd rasteriser needs CUDA/HIP tensors (no CPU fallback)")
    rs = _RS(raster_settings, device)
    P, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
    color = torch.empty((K, 3, H, W), dtype=torch.float32, device=device)
    depth = torch.empty((K, 1, H, W), dtype=torch.float32, device=device)
    radii = torch.empty((K, P), dtype=torch.int32, device=device)
    geom = torch.empty(L.dgs_geom_state_bytes(P, K), dtype=torch.uint8, device=device)
    image = torch.empty(L.dgs_image_state_bytes_forward_only(W, H, K) if forward_only else
                        L.dgs_image_state_bytes(W, H, K), dtype=torch.uint8, device=device)
    host_R = _pinned_word(device)
    out = _lib.DgsForwardOut()
    out.out_color = _ptr(color)
    out.out_depth = _ptr(depth)
    out.radii = _ptr(radii)
    out.num_rendered_host = ctypes.c_void_p(host_R.data_ptr())
    chk = torch.zeros((K, H * W), dtype=torch.int32, device=device) if debug_checksum else None
    out.debug_contrib_checksum = _ptr(chk)
    stream = _stream(device)
    tile_cull = bool(TILE_CULL)
    prob = _make_problem(K, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewm, projm,
                         campos, rs, geom, image, None, tile_cull, raw)
    prob.forward_only = int(bool(forward_only))
    if raw is not None:
        prob.M = 1 + (0 if raw["sh_rest"] is None else raw["sh_rest"].shape[1])
    if capacity is not None:
        binning = torch.empty(L.dgs_binning_state_bytes(int(capacity), W, H, K), dtype=torch.uint8, device=device)
        prob.binning_state = _ptr(binning)
        prob.binning_bytes = binning.numel()
        _lib.check(L.dgs_forward(ctypes.byref(prob), ctypes.byref(out), int(capacity), stream), "dgs_forward")
        torch.cuda.current_stream(device).synchronize()
        R = _NumRendered(int(host_R[3].item()) & 0xFFFFFFFF)
        R.tile_cull, R.capacity, R.overflow = tile_cull, int(capacity), bool(int(host_R[2].item()))
        R.counted = int(host_R[0].item()) & 0xFFFFFFFF
        return R, color, depth, radii, geom, binning, image
    _lib.check(L.dgs_forward_geometry(ctypes.byref(prob), ctypes.byref(out), stream), "dgs_forward_geometry")
    torch.cuda.current_stream(device).synchronize()   # the one host read of num_rendered (rasterizer_impl.cu:287)
    if int(host_R[1].item()) != 0:
        raise RuntimeError("num_rendered exceeds 32 bits: too many (tile, Gaussian) duplicates for one fused call; "
                           "render fewer subframes per call")
    R = _NumRendered(int(host_R[0].item()) & 0xFFFFFFFF)
    R.tile_cull = tile_cull
    binning = torch.empty(L.dgs_binning_state_bytes(R, W, H, K), dtype=torch.uint8, device=device)
    prob.binning_state = _ptr(binning)
    prob.binning_bytes = binning.numel()
    _lib.check(L.dgs_forward_render(ctypes.byref(prob), ctypes.byref(out), R, stream), "dgs_forward_render")
    R.contrib_checksum = chk
    return R, color, depth, radii, geom, binning, image


def _backward_impl(K, R, means3D, sh, colors_precomp, opacities_shape, scales, rotations, cov3D_precomp, viewm, projm,
                   campos, raster_settings, radii, geom, binning, image, grad_color, grad_depth):
    L = _lib.lib()
    device = means3D.device
    rs = _RS(raster_settings, device)
    P = means3D.shape[0]
    M = 0 if sh is None else sh.shape[1]
    f = dict(dtype=torch.float32, device=device)
    g_means3D = torch.empty((P, 3), **f)
    g_means2D = torch.empty((K, P, 3), **f)
    g_sh = torch.empty((P, M, 3), **f) if sh is not None else None
    g_colors = torch.empty((P, 3), **f)
    g_opacity = torch.empty((P, 1), **f)
    g_scales = torch.empty((P, 3), **f) if scales is not None else None
    g_rots = torch.empty((P, 4), **f) if rotations is not None else None
    g_cov3D = torch.empty((P, 6), **f)
    g_view = torch.empty((K, 4, 4), **f)
    g_proj = torch.empty((K, 4, 4), **f)
    scratch = torch.empty(L.dgs_backward_scratch_bytes(R, P, K), dtype=torch.uint8, device=device)
    io = _lib.DgsBackwardIO()
    io.num_rendered = R
    io.radii = _ptr(radii)
    io.dL_dout_color = _ptr(grad_color)
    io.dL_dout_depth = _ptr(grad_depth)
    io.scratch = _ptr(scratch)
    io.scratch_bytes = scratch.numel()
    io.dL_dmeans3D = _ptr(g_means3D)
    io.dL_dmeans2D = _ptr(g_means2D)
    io.dL_dsh = _ptr(g_sh)
    io.dL_dcolors = _ptr(g_colors)
    io.dL_dopacity = _ptr(g_opacity)
    io.dL_dscales = _ptr(g_scales)
    io.dL_drotations = _ptr(g_rots)
    io.dL_dcov3D = _ptr(g_cov3D)
    io.dL_dviewmatrix = _ptr(g_view)
    io.dL_dprojmatrix = _ptr(g_proj)
    prob = _make_problem(K, means3D, sh, colors_precomp, None, scales, rotations, cov3D_precomp, viewm, projm, campos,
                         rs, geom, image, binning, getattr(R, "tile_cull", False))
    _lib.check(L.dgs_backward(ctypes.byref(prob), ctypes.byref(io), _stream(device)), "dgs_backward")
    if BACKWARD_DEBUG is not None:
        BACKWARD_DEBUG.update(scratch=scratch, R=int(R), K=K, P=P, dL_dcov3D=g_cov3D, dL_dcolors=g_colors)
    if P == 0:
        for t in (g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rots, g_cov3D):
            if t is not None:
                t.zero_()
    return g_means2D, g_colors, g_opacity, g_means3D, g_cov3D, g_sh, g_scales, g_rots, g_view, g_proj


def _prep(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp):
    return (_f32c(means3D), _f32c(_opt(sh)), _f32c(_opt(colors_precomp)), _f32c(opacities), _f32c(_opt(scales)),
            _f32c(_opt(rotations)), _f32c(_opt(cov3Ds_precomp)))


# The reference's second use of the operator is inference: test.py:117 and render_spiral.py:29 call render() under
# torch.no_grad().  When no input can receive a gradient the entry points below skip the autograd Function and run the
# forward with DgsProblem.forward_only = 1 (nothing stored for a backward; same images, same radii).
FORWARD_ONLY_WHEN_NO_GRAD = True


def _inference(*tensors):
    return FORWARD_ONLY_WHEN_NO_GRAD and not (torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in tensors))


# ------------------------------------------------------------------------------------- K = 1 (reference API)
def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        viewmatrix, projmatrix, raster_settings):
    if _inference(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                  projmatrix) and not raster_settings.debug:
        m3, shc, colc, opc, scc, rotc, covc = _prep(means3D, sh, colors_precomp, opacities, scales, rotations,
                                                    cov3Ds_precomp)
        with torch.no_grad():
            _, color, depth, radii, _, _, _ = _forward_impl(
                1, m3, shc, colc, opc, scc, rotc, covc, _f32c(viewmatrix).reshape(1, 4, 4),
                _f32c(projmatrix).reshape(1, 4, 4), _f32c(raster_settings.campos.to(m3.device)).reshape(1, 3),
                raster_settings, forward_only=True)
        return color[0], depth[0], radii[0]
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, viewmatrix, projmatrix, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                projmatrix, raster_settings):
        m3, shc, colc, opc, scc, rotc, covc = _prep(means3D, sh, colors_precomp, opacities, scales, rotations,
                                                    cov3Ds_precomp)
        viewm = _f32c(viewmatrix).reshape(1, 4, 4)
        projm = _f32c(projmatrix).reshape(1, 4, 4)
        campos = _f32c(raster_settings.campos.to(m3.device)).reshape(1, 3)
        args = (m3, shc, colc, opc, scc, rotc, covc, viewm, projm, campos, raster_settings)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args[:-1])  # copy them before they can be corrupted
            try:
                num_rendered, color, depth, radii, geom, binning, img = _forward_impl(1, *args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, depth, radii, geom, binning, img = _forward_impl(1, *args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.opacities_shape = opacities.shape
        ctx.absent = (shc is None, colc is None, scc is None, rotc is None, covc is None)
        dummy = m3.new_empty(0)
        ctx.save_for_backward(*(dummy if t is None else t for t in (colc, m3, scc, rotc, covc, radii, shc, geom,
                                                                    binning, img, viewm, projm, campos)))
        ctx.set_materialize_grads(False)
        color, depth, radii = color[0], depth[0], radii[0]
        ctx.mark_non_differentiable(radii)
        return color, depth, radii

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth, _):
        rs = ctx.raster_settings
        colc, m3, scc, rotc, covc, radii, shc, geom, binning, img, viewm, projm, campos = (
            None if t.numel() == 0 and i != 1 else t for i, t in enumerate(ctx.saved_tensors))
        if grad_out_color is None and grad_out_depth is None:
            return (None,) * 11
        H, W = int(rs.image_height), int(rs.image_width)
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, H, W), dtype=torch.float32, device=m3.device)
        gc = _f32c(grad_out_color)
        gd = _f32c(grad_out_depth)
        args = (1, ctx.num_rendered, m3, shc, colc, ctx.opacities_shape, scc, rotc, covc, viewm, projm, campos, rs,
                radii, geom, binning, img, gc, gd)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                grads = _backward_impl(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            grads = _backward_impl(*args)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_viewmatrix, grad_projmatrix) = grads
        return (
            grad_means3D,
            grad_means2D[0],
            grad_sh,
            grad_colors_precomp if colc is not None else None,
            grad_opacities.reshape(ctx.opacities_shape),
            grad_scales,
            grad_rotations,
            grad_cov3Ds_precomp if covc is not None else None,
            grad_viewmatrix[0],
            grad_projmatrix[0],
            None,
        )


# ------------------------------------------------------------------------------ K subframes, one fused launch
def rasterize_gaussians_subframes(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                  viewmatrices, projmatrices, raster_settings):
    if _inference(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrices,
                  projmatrices) and not raster_settings.debug:
        m3, shc, colc, opc, scc, rotc, covc = _prep(means3D, sh, colors_precomp, opacities, scales, rotations,
                                                    cov3Ds_precomp)
        viewm, projm = _f32c(viewmatrices), _f32c(projmatrices)
        K = viewm.shape[0]
        campos = _f32c(raster_settings.campos.to(m3.device)).reshape(-1, 3)
        if viewm.shape != (K, 4, 4) or projm.shape != (K, 4, 4) or campos.shape[0] != K:
            raise RuntimeError("viewmatrices / projmatrices must be [K,4,4] and raster_settings.campos [K,3]")
        with torch.no_grad():
            _, color, depth, radii, _, _, _ = _forward_impl(K, m3, shc, colc, opc, scc, rotc, covc, viewm, projm, campos,
                                                            raster_settings, forward_only=True)
        return color, depth, radii
    return _RasterizeGaussiansK.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                      cov3Ds_precomp, viewmatrices, projmatrices, raster_settings)


class _RasterizeGaussiansK(torch.autograd.Function):
    """All K subframes of one blurry view: viewmatrices/projmatrices [K,4,4], raster_settings.campos [K,3],
    means2D [K,P,3] (its .grad carries the per-subframe screen-space gradients that densification reads,
    train.py:188-193).  Returns color [K,3,H,W], depth [K,1,H,W], radii [K,P]."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                viewmatrices, projmatrices, raster_settings):
        m3, shc, colc, opc, scc, rotc, covc = _prep(means3D, sh, colors_precomp, opacities, scales, rotations,
                                                    cov3Ds_precomp)
        viewm = _f32c(viewmatrices)
        projm = _f32c(projmatrices)
        K = viewm.shape[0]
        if viewm.shape != (K, 4, 4) or projm.shape != (K, 4, 4):
            raise RuntimeError("viewmatrices / projmatrices must be [K,4,4]")
        campos = _f32c(raster_settings.campos.to(m3.device)).reshape(-1, 3)
        if campos.shape[0] != K:
            raise RuntimeError("raster_settings.campos must be [K,3] for the K-subframe operator")
        num_rendered, color, depth, radii, geom, binning, img = _forward_impl(
            K, m3, shc, colc, opc, scc, rotc, covc, viewm, projm, campos, raster_settings)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.K = K
        ctx.opacities_shape = opacities.shape
        dummy = m3.new_empty(0)
        ctx.save_for_backward(*(dummy if t is None else t for t in (colc, m3, scc, rotc, covc, radii, shc, geom,
                                                                    binning, img, viewm, projm, campos)))
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        return color, depth, radii

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth, _):
        rs = ctx.raster_settings
        K = ctx.K
        colc, m3, scc, rotc, covc, radii, shc, geom, binning, img, viewm, projm, campos = (
            None if t.numel() == 0 and i != 1 else t for i, t in enumerate(ctx.saved_tensors))
        if grad_out_color is None and grad_out_depth is None:
            return (None,) * 11
        H, W = int(rs.image_height), int(rs.image_width)
        if grad_out_color is None:
            grad_out_color = torch.zeros((K, 3, H, W), dtype=torch.float32, device=m3.device)
        grads = _backward_impl(K, ctx.num_rendered, m3, shc, colc, ctx.opacities_shape, scc, rotc, covc, viewm, projm,
                               campos, rs, radii, geom, binning, img, _f32c(grad_out_color), _f32c(grad_out_depth))
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_viewmatrix, grad_projmatrix) = grads
        return (
            grad_means3D,
            grad_means2D,
            grad_sh,
            grad_colors_precomp if colc is not None else None,
            grad_opacities.reshape(ctx.opacities_shape),
            grad_scales,
            grad_rotations,
            grad_cov3Ds_precomp if covc is not None else None,
            grad_viewmatrix,
            grad_projmatrix,
            None,
        )


# ------------------------------------------------ K subframes straight from the cloud's raw parameters
def _align4(n):
    return (n + 3) // 4 * 4


class _RasterizeCloudK(torch.autograd.Function):
    """The K-subframe operator with the reference's parameter activations folded into the kernels
    (DgsProblem.raw_params): inputs are GaussianModel's raw tensors (_xyz, _features_dc [P,1,3], _features_rest
    [P,M-1,3], _opacity, _scaling, _rotation) instead of the activated values render() computes with
    get_opacity / get_scaling / get_rotation / get_features (gaussian_renderer/__init__.py:60-77,
    scene/gaussian_model.py:114-137), so the ~25 elementwise / cat / norm launches of those getters and of their
    autograd backward disappear.  The six gradients are views of ONE flat buffer, in the order of the reference's
    optimiser groups, so a data-parallel run can all-reduce them without packing (sharding.flat_allreduce_grads)."""

    @staticmethod
    def forward(ctx, xyz, means2D, f_dc, f_rest, opacity, scaling, rotation, viewmatrices, projmatrices,
                raster_settings, scale_lb, isotropic=False):
        m3, dc, opc, scc, rotc = (_f32c(t) for t in (xyz, f_dc, opacity, scaling, rotation))
        rest = _f32c(f_rest) if f_rest is not None and f_rest.shape[1] > 0 else None
        viewm, projm = _f32c(viewmatrices), _f32c(projmatrices)
        K = viewm.shape[0]
        campos = _f32c(raster_settings.campos.to(m3.device)).reshape(-1, 3)
        if viewm.shape != (K, 4, 4) or projm.shape != (K, 4, 4) or campos.shape[0] != K:
            raise RuntimeError("viewmatrices / projmatrices must be [K,4,4] and raster_settings.campos [K,3]")
        raw = {"scale_lb": float(scale_lb), "sh_rest": rest, "isotropic": bool(isotropic)}
        ctx.isotropic = bool(isotropic)
        num_rendered, color, depth, radii, geom, binning, img = _forward_impl(
            K, m3, dc.reshape(-1, 1, 3), None, opc.reshape(-1), scc, rotc, None, viewm, projm, campos, raster_settings,
            raw=raw)
        ctx.raster_settings, ctx.num_rendered, ctx.K, ctx.scale_lb = raster_settings, num_rendered, K, float(scale_lb)
        ctx.shapes = (f_dc.shape, None if f_rest is None else f_rest.shape, opacity.shape)
        dummy = m3.new_empty(0)
        ctx.save_for_backward(m3, dc, dummy if rest is None else rest, opc, scc, rotc, radii, geom, binning, img, viewm,
                              projm, campos)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        return color, depth, radii

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth, _):
        rs, K = ctx.raster_settings, ctx.K
        m3, dc, rest, opc, scc, rotc, radii, geom, binning, img, viewm, projm, campos = ctx.saved_tensors
        rest = None if rest.numel() == 0 else rest
        if grad_out_color is None and grad_out_depth is None:
            return (None,) * 12
        H, W = int(rs.image_height), int(rs.image_width)
        device = m3.device
        if grad_out_color is None:
            grad_out_color = torch.zeros((K, 3, H, W), dtype=torch.float32, device=device)
        gc, gd = _f32c(grad_out_color), _f32c(grad_out_depth)
        L = _lib.lib()
        R = ctx.num_rendered
        P = m3.shape[0]
        Mr = 0 if rest is None else rest.shape[1]
        f = dict(dtype=torch.float32, device=device)
        # one flat gradient buffer, segments 16-byte aligned, in optimiser-group order
        sizes = [3 * P, 3 * P, 3 * Mr * P, P, 3 * P, 4 * P]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + _align4(n))
        flat = torch.empty(offs[-1], **f)
        seg = lambda i, shape: flat[offs[i]:offs[i] + sizes[i]].view(shape)
        g_xyz, g_dc, g_op, g_sc, g_rot = seg(0, (P, 3)), seg(1, (P, 1, 3)), seg(3, (P, 1)), seg(4, (P, 3)), seg(5, (P, 4))
        g_rest = seg(2, (P, Mr, 3)) if Mr > 0 else None
        g_means2D = torch.empty((K, P, 3), **f)
        g_colors = torch.empty((P, 3), **f)
        g_cov3D = torch.empty((P, 6), **f)
        g_view, g_proj = torch.empty((K, 4, 4), **f), torch.empty((K, 4, 4), **f)
        scratch = torch.empty(L.dgs_backward_scratch_bytes(R, P, K), dtype=torch.uint8, device=device)
        io = _lib.DgsBackwardIO()
        io.num_rendered = R
        io.radii, io.dL_dout_color, io.dL_dout_depth = _ptr(radii), _ptr(gc), _ptr(gd)
        io.scratch, io.scratch_bytes = _ptr(scratch), scratch.numel()
        io.dL_dmeans3D, io.dL_dmeans2D, io.dL_dsh, io.dL_dsh_rest = _ptr(g_xyz), _ptr(g_means2D), _ptr(g_dc), _ptr(g_rest)
        io.dL_dcolors, io.dL_dopacity, io.dL_dscales, io.dL_drotations = _ptr(g_colors), _ptr(g_op), _ptr(g_sc), _ptr(g_rot)
        io.dL_dcov3D, io.dL_dviewmatrix, io.dL_dprojmatrix = _ptr(g_cov3D), _ptr(g_view), _ptr(g_proj)
        prob = _make_problem(K, m3, dc, None, opc.reshape(-1), scc, rotc, None, viewm, projm, campos, _RS(rs, device),
                             geom, img, binning, getattr(R, "tile_cull", False),
                             raw={"scale_lb": ctx.scale_lb, "sh_rest": rest, "isotropic": ctx.isotropic})
        prob.M = 1 + Mr
        _lib.check(L.dgs_backward(ctypes.byref(prob), ctypes.byref(io), _stream(device)), "dgs_backward")
        if BACKWARD_DEBUG is not None:
            BACKWARD_DEBUG.update(scratch=scratch, R=int(R), K=K, P=P, dL_dcov3D=g_cov3D, dL_dcolors=g_colors)
        if P == 0:
            flat.zero_()
            g_means2D.zero_()
        dc_shape, rest_shape, op_shape = ctx.shapes
        g_rest_out = g_rest.view(rest_shape) if g_rest is not None else (
            None if rest_shape is None else flat.new_empty(rest_shape))
        return (g_xyz, g_means2D, g_dc.view(dc_shape), g_rest_out,
                g_op.view(op_shape), g_sc, g_rot, g_view, g_proj, None, None, None)


def rasterize_cloud_subframes(xyz, means2D, f_dc, f_rest, opacity, scaling, rotation, viewmatrices, projmatrices,
                              raster_settings, scale_lb=0.0, isotropic=False):
    """isotropic: the cloud has one shared scale per Gaussian, column 0 of `scaling` (use_isotrophic,
    scene/gaussian_model.py:115-118); the gradient of `scaling` then has zeros in columns 1 and 2."""
    if _inference(xyz, means2D, f_dc, f_rest, opacity, scaling, rotation, viewmatrices, projmatrices):
        m3, dc, opc, scc, rotc = (_f32c(t) for t in (xyz, f_dc, opacity, scaling, rotation))
        rest = _f32c(f_rest) if f_rest is not None and f_rest.shape[1] > 0 else None
        viewm, projm = _f32c(viewmatrices), _f32c(projmatrices)
        K = viewm.shape[0]
        campos = _f32c(raster_settings.campos.to(m3.device)).reshape(-1, 3)
        if viewm.shape != (K, 4, 4) or projm.shape != (K, 4, 4) or campos.shape[0] != K:
            raise RuntimeError("viewmatrices / projmatrices must be [K,4,4] and raster_settings.campos [K,3]")
        with torch.no_grad():
            _, color, depth, radii, _, _, _ = _forward_impl(
                K, m3, dc.reshape(-1, 1, 3), None, opc.reshape(-1), scc, rotc, None, viewm, projm, campos, raster_settings,
                raw={"scale_lb": float(scale_lb), "sh_rest": rest, "isotropic": bool(isotropic)}, forward_only=True)
        return color, depth, radii
    return _RasterizeCloudK.apply(xyz, means2D, f_dc, f_rest, opacity, scaling, rotation, viewmatrices, projmatrices,
                                  raster_settings, scale_lb, bool(isotropic))


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions, viewmatrix=None, projmatrix=None):
        """Frustum-visibility mask.  In the reference fork this method reads raster_settings.viewmatrix, a field
        that no longer exists (dead code, __init__.py:194-203); here the matrices are explicit arguments."""
        with torch.no_grad():
            positions = _f32c(positions)
            vm = _f32c(viewmatrix if viewmatrix is not None else getattr(self.raster_settings, "viewmatrix"))
            pm = _f32c(projmatrix if projmatrix is not None else getattr(self.raster_settings, "projmatrix", vm))
            visible = torch.empty(positions.shape[0], dtype=torch.bool, device=positions.device)
            _lib.check(_lib.lib().dgs_mark_visible(positions.shape[0], _ptr(positions), _ptr(vm), _ptr(pm),
                                                   _ptr(visible), _stream(positions.device)), "dgs_mark_visible")
        return visible

    @staticmethod
    def _check(shs, colors_precomp, scales, rotations, cov3D_precomp):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, viewmatrix=None, projmatrix=None):
        raster_settings = self.raster_settings
        self._check(shs, colors_precomp, scales, rotations, cov3D_precomp)
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, viewmatrix, projmatrix, raster_settings)

    def forward_subframes(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                          rotations=None, cov3D_precomp=None, viewmatrices=None, projmatrices=None):
        """K-subframe sibling of forward(): matrices are [K,4,4], raster_settings.campos is [K,3]."""
        self._check(shs, colors_precomp, scales, rotations, cov3D_precomp)
        e = torch.Tensor([])
        return rasterize_gaussians_subframes(
            means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp, opacities,
            e if scales is None else scales, e if rotations is None else rotations,
            e if cov3D_precomp is None else cov3D_precomp, viewmatrices, projmatrices, self.raster_settings)
