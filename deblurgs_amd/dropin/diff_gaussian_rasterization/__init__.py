"""Drop-in shim: put `deblurgs_amd/dropin` on PYTHONPATH ahead of the reference's CUDA extension and its
`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(gaussian_renderer/__init__.py:14) resolves to the MI355X operator.  See INTEGRATION.md."""
from deblurgs_amd.diff_gaussian_rasterization import (  # noqa: F401
    GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians, _RasterizeGaussians,
    rasterize_gaussians_subframes, _RasterizeGaussiansK)
