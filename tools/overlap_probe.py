"""Measuring tool: how much of the forward's HBM-bound binning hides under another subframe group's VALU-bound
compositing when two groups of subframes are enqueued on two HIP streams (estimate for DESIGN section 9)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import synthetic, hip_settings, _t
from deblurgs_amd import diff_gaussian_rasterization as dgr
from deblurgs_amd.cloud import GaussianCloud
sc = synthetic.make_config("metric"); K = sc["K"]
c = GaussianCloud.from_scene(sc, "cuda")
view, proj, cam = _t(sc["viewmatrix"]), _t(sc["projmatrix"]), _t(sc["campos"])
raw = {"scale_lb": 0.0, "sh_rest": c._features_rest}
def fwd(k0, k1):
    rs = hip_settings(sc, k1 - k0)._replace(campos=cam[k0:k1])
    with torch.no_grad():
        return dgr._forward_impl(k1 - k0, c._xyz, c._features_dc, None, c._opacity.reshape(-1), c._scaling, c._rotation, None,
                                 view[k0:k1].contiguous(), proj[k0:k1].contiguous(), cam[k0:k1].contiguous(), rs, raw=raw)
def timeit(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both_serial():
    fwd(0, 8); fwd(8, 15)
def both_streams():
    # NB: the two-phase forward blocks on each stream's count; the groups still interleave on the device
    import threading
    def run(s, a, b):
        with torch.cuda.stream(s):
            fwd(a, b)
    t1 = threading.Thread(target=run, args=(s1, 0, 8)); t2 = threading.Thread(target=run, args=(s2, 8, 15))
    t1.start(); t2.start(); t1.join(); t2.join()
print("one call K=15      %.2f ms" % timeit(lambda: fwd(0, 15)))
print("two calls serial   %.2f ms" % timeit(both_serial))
print("two calls 2 streams %.2f ms" % timeit(both_streams))
