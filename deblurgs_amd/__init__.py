"""deblurgs_amd -- MI355X-native differentiable Gaussian-splat rasteriser for DeblurGS's blur-integration loop.

Only the hot path of taekkii/deblurgs lives here: the rasteriser operator surface
(`diff_gaussian_rasterization`), the render adapter (`gaussian_renderer`), the subframe loop
(`motion.CameraMotionModule.query`), the pose path (`pose`), the loss block (`losses`) and the subframe/view
sharding over GPUs (`sharding`).  All device work goes through the C ABI of ``libdgs_hip.so``
(include/dgs_hip.h); there is no CPU fallback -- importing the kernels without the built library raises.
"""
__version__ = "0.1.0"
