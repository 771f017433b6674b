"""Render adapter: same signature and return dict as the reference's gaussian_renderer.render()
(/root/reference/gaussian_renderer/__init__.py:18-90), plus the batched sibling render_subframes() that the
blur-integration loop uses (one fused launch chain for the K subframe cameras of a blurry view instead of K
render() calls, scene/motion.py:141-143).

`pc` is any object exposing the attributes the reference's GaussianModel getters expose
(scene/gaussian_model.py:114-137): get_xyz, get_opacity, get_scaling, get_rotation, get_features,
active_sh_degree, z_near, z_far, use_sigmoid.
"""
import math

import torch

from .diff_gaussian_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,
                                          rasterize_cloud_subframes)


def render(viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render the scene.  Background tensor (bg_color) must be on the GPU."""
    # zero tensor whose .grad receives the 2-D (screen-space) mean gradients
    if torch.is_grad_enabled():
        screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True,
                                              device=pc.get_xyz.device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    else:   # inference: nothing will ever be written to its .grad; the dict keeps the reference's zeros tensor
        screenspace_points = torch.zeros_like(pc.get_xyz)
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        z_near=pc.z_near,
        z_far=pc.z_far,
        use_sigmoid=pc.use_sigmoid,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=False,
    )
    if override_color is None and getattr(pc, "fused_activations", False) and not torch.is_grad_enabled():
        # inference on a cloud whose activations the kernels apply themselves (as render_subframes does): no getter
        # launches, no dc | rest concat (108 MB per frame at 1 M Gaussians, SH degree 2) -- same images, same radii
        raster_settings = raster_settings._replace(campos=viewpoint_camera.camera_center.reshape(1, 3))
        images, depths, radii = rasterize_cloud_subframes(
            pc._xyz, None, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation,
            viewpoint_camera.world_view_transform.reshape(1, 4, 4), viewpoint_camera.full_proj_transform.reshape(1, 4, 4),
            raster_settings, pc.scale_lower_bound, isotropic=getattr(pc, "use_isotrophic", False))
        return {"render": images[0], "depth": depths[0], "viewspace_points": screenspace_points,
                "visibility_filter": radii[0] > 0, "radii": radii[0]}
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    shs = None
    colors_precomp = None
    if override_color is None:
        shs = pc.get_features
    else:
        colors_precomp = override_color
    rendered_image, rendered_depth, radii = rasterizer(
        means3D=pc.get_xyz,
        means2D=screenspace_points,
        shs=shs,
        colors_precomp=colors_precomp,
        opacities=pc.get_opacity,
        scales=pc.get_scaling,
        rotations=pc.get_rotation,
        cov3D_precomp=None,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform)
    return {"render": rendered_image,
            "depth": rendered_depth,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii}


def render_subframes(world_views, full_projs, camera_centers, ref_camera, pc, bg_color: torch.Tensor,
                     scaling_modifier=1.0, override_color=None):
    """All K subframes of one blurry view in one fused launch.

    world_views / full_projs: [K,4,4] (may require grad: the trajectory is optimised through them);
    camera_centers: [K,3]; ref_camera supplies image size and FoV (scene/motion.py:283-292).
    The activation getters and the SH concat are evaluated ONCE, not K times.
    Returns the stacked equivalents of K render() dicts:
      render [K,3,H,W], depth [K,1,H,W], viewspace_points [K,P,3] (grad carrier), visibility_filter [K,P],
      radii [K,P]."""
    xyz = pc.get_xyz
    K = world_views.shape[0]
    # a leaf whose .grad receives the per-subframe screen-space gradients (the reference's `zeros + 0` +
    # retain_grad() idiom costs an extra elementwise pass over [K,P,3] for the same effect)
    # (its values are never read -- the rasteriser ignores means2D, forward.cu has no use for it -- so it is not
    # zero-filled: at the metric config that is 180 MB per step)
    screenspace_points = torch.empty((K,) + tuple(xyz.shape), dtype=xyz.dtype, device=xyz.device).requires_grad_(True)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(ref_camera.image_height),
        image_width=int(ref_camera.image_width),
        tanfovx=math.tan(ref_camera.FoVx * 0.5),
        tanfovy=math.tan(ref_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        z_near=pc.z_near,
        z_far=pc.z_far,
        use_sigmoid=pc.use_sigmoid,
        sh_degree=pc.active_sh_degree,
        campos=camera_centers,
        prefiltered=False,
        debug=False,
    )
    if override_color is None and getattr(pc, "fused_activations", False):
        # the cloud's activations (opacity clamp, exp(+lb) scale, quaternion normalise, dc|rest concat) run inside
        # the kernels, forward and backward: no getter launches, gradients land on the raw parameters directly
        images, depths, radii = rasterize_cloud_subframes(
            pc._xyz, screenspace_points, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation,
            world_views, full_projs, raster_settings, pc.scale_lower_bound,
            isotropic=getattr(pc, "use_isotrophic", False))
        return {"render": images, "depth": depths, "viewspace_points": screenspace_points,
                "visibility_filter": radii > 0, "radii": radii}
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    shs = None
    colors_precomp = None
    if override_color is None:
        shs = pc.get_features
    else:
        colors_precomp = override_color
    images, depths, radii = rasterizer.forward_subframes(
        means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp, opacities=pc.get_opacity,
        scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=None, viewmatrices=world_views,
        projmatrices=full_projs)
    return {"render": images,
            "depth": depths,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii}
