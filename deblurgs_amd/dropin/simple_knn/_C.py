"""Drop-in shim: with `deblurgs_amd/dropin` on PYTHONPATH the reference's `from simple_knn._C import distCUDA2`
(scene/gaussian_model.py:20) resolves to the MI355X kernel.  See INTEGRATION.md."""
from deblurgs_amd.simple_knn import distCUDA2  # noqa: F401
