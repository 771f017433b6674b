import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_train import _fused_fixture
from deblurgs_amd import losses
from deblurgs_amd.fused_step import FusedStep
sc, cloud, m = _fused_fixture()
bg = torch.tensor([0.2, 0.5, 0.1], device="cuda")
out = m.query(1, "all", background=bg, compute_blurred=False)
fs = FusedStep(cloud, m, lambda_hinge=0.1, speculative=False)
fr = fs.run(1, 2e-3, m.get_gt_image(1), bg, "all", need_blur=True)
torch.cuda.synchronize()
print("last_capacity", fs.last_capacity, "host", fs._pending[0].host.tolist())
print("img diff", float((fr["subframes"] - out["subframes"]).abs().max()), "radii diff", int((fr["radii"] != out["radii_all"]).sum()))
nu_t = m._sample_nu_from_alignment(1)
print("nu torch", nu_t.tolist())
print("nu kernel", fs._keep[10].tolist())
wv, fp, cc = m.get_trajectory_matrices(1)
print("view diff", float((wv - fs._keep[7]).abs().max()), float((fp - fs._keep[8]).abs().max()))
